#!/usr/bin/env python
"""bench.py — read-alignments/s of the genotyping hot path on N MI355X GPUs.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it
is launched under torch.distributed.run, one rank per GPU over RCCL.  One JSON
line is printed by rank 0.

A "step" = one pass of the hot path (vtx_run: Smith-Waterman of every record
against both haplotypes, per-read call, UMI collapse, per-(row, cell) histogram,
ordered COO emit) over one resident synthetic batch, plus — for N > 1 — the
gather of every rank's matrix rows to rank 0.  Inputs are resident in HBM
before the timed region (vtx_submit is outside it).  Workload at N = 1:
BASELINE.json configs[2], "synthetic 100k SNV loci x 10k barcodes, consensus
mode" (the configuration the metric is quoted on); each extra GPU adds one more
such shard of loci (weak scaling; rows of rank r are offset by r * n_loci).

metric value = read-vs-haplotype alignments (2 per scored read) of all ranks /
max-over-ranks wall time of the K timed steps.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from vartrix_amd import lib, shard, synth  # noqa: E402
from vartrix_amd.abi import default_config  # noqa: E402

# Algorithmic HBM bytes per alignment (SURVEY.md §8d / BASELINE.md §3): per scored
# read 150 B bases + 12 B record + 8 B scores out, over 2 alignments, + the locus'
# haplotypes and descriptor amortised over its reads.
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_PEAK_TOPS = 39.3          # packed 16-bit VALU ops issue at 4 cycles / wave64 on gfx950: 256 CU x 4 SIMD x 16
                               # lanes/clk x 2.4 GHz (measured 38.7, profiles/r01_valu_peak_microbench.txt)
OPS_PER_CELL_PAIR = 9          # packed VALU ops per DP cell pair of sw_full_lut_kernel; the
                               # byte-equality fallback sw_full_kernel spends 12


def algorithmic_bytes(batch) -> int:
    rec = batch.records
    loci = batch.loci
    per_read = int(rec["read_len"].astype(np.int64).sum()) + 12 * batch.n_records + 8 * batch.n_records
    per_locus = int((loci["ref_len"].astype(np.int64) + loci["alt_len"]).sum()) + 32 * batch.n_loci
    return per_read + per_locus


def cpu_baseline(batch, cfg, target_seconds=15.0):
    """Oracle ("port" of the reference CPU path, static chunks over all host cores) on a
    bounded sample of the same workload."""
    from oracle import oracle   # test infrastructure: imported here only, as the reported baseline
    cores = os.cpu_count() or 1
    probe = batch.slice_loci(0, min(batch.n_loci, max(cores, 8)))
    t0 = time.perf_counter()
    oracle.batch_scores(probe, cfg, threads=cores)
    dt = max(time.perf_counter() - t0, 1e-3)
    rate = 2 * probe.n_records / dt
    n_loci = int(min(batch.n_loci, max(probe.n_loci, target_seconds * rate / 2 / max(batch.n_records / batch.n_loci, 1))))
    sample = batch.slice_loci(0, n_loci)
    t0 = time.perf_counter()
    ref, alt = oracle.batch_scores(sample, cfg, threads=cores)
    oracle.batch_reduce(sample, cfg, ref, alt)
    dt = time.perf_counter() - t0
    return {"value": 2 * sample.n_records / dt, "unit": "read-alignments/s", "cores": cores, "kind": "port",
            "sample": "first %d loci (%d scored reads, %.1f s) of the same batch, %s aligner, %d OpenMP threads"
                      % (sample.n_loci, sample.n_records, dt, "full" if cfg.aligner == 1 else "banded", cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--loci", type=int, default=100_000)
    ap.add_argument("--barcodes", type=int, default=10_000)
    ap.add_argument("--reads-per-locus", type=int, default=256)
    ap.add_argument("--mode", default="consensus", choices=["consensus", "alt_frac", "coverage"])
    ap.add_argument("--aligner", default="banded", choices=["banded", "full"],
                    help="banded = the reference's banded::Aligner semantics (default); full = unbanded Smith-Waterman")
    ap.add_argument("--umi", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-aligner", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    force_gather = os.environ.get("VTX_FORCE_GATHER") == "1"     # exercise the RCCL path with one rank
    if world > 1 or force_gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    spec = synth.SynthSpec(n_loci=args.loci, n_barcodes=args.barcodes, reads_per_locus=args.reads_per_locus,
                           use_umi=bool(args.umi), seed=20260926 + rank)
    t_gen = time.perf_counter()
    batch = synth.make_batch(spec)
    batch.loci["row"] += np.uint32(rank * args.loci)      # this rank's shard of matrix rows
    t_gen = time.perf_counter() - t_gen
    cfg = default_config(aligner=args.aligner, scoring_mode=args.mode, use_umi=args.umi,
                         n_barcodes=args.barcodes, device=local_rank)
    ctx = lib.Context(cfg)
    t_sub = time.perf_counter()
    ctx.submit(batch)                                      # H2D: outside the timed region
    t_sub = time.perf_counter() - t_sub

    pending = [None]      # in-flight row gather of the previous step (overlaps with this step's kernels)

    def step():
        ctx.run()
        if world > 1 or force_gather:
            local = shard.device_coo_tensors(ctx, device)
            handle = shard.gather_coo_async(local, cfg.scoring_mode)   # packs a copy, then async gather
            if pending[0] is not None:
                pending[0].wait()
            pending[0] = handle

    def drain():
        if pending[0] is not None:
            pending[0].wait()
            pending[0] = None

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    sw_ms, red_ms, full_ms, band_ms = [], [], [], []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        t = ctx.timing()                                   # hipEvents on the context's own stream
        sw_ms.append(t.sw_ms)
        red_ms.append(t.reduce_ms)
        full_ms.append(t.full_ms)
        band_ms.append(t.band_ms)
    drain()               # every step's gather has completed inside the timed region
    fence()
    elapsed = time.perf_counter() - t0
    n_aln = 2 * batch.n_records
    cells = ctx.cells()
    nnz = ctx.device_coo()["nnz"]

    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        tot = torch.tensor([n_aln, cells, nnz], dtype=torch.int64, device=device)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_aln, total_cells, total_nnz = (int(x) for x in tot.tolist())
    else:
        total_aln, total_cells, total_nnz = n_aln, cells, nnz

    # secondary: the other aligner flavour on the same resident-size batch (rank 0 only, single GPU timing)
    other = None
    if rank == 0 and world == 1 and not args.no_other_aligner:      # N = 1 only: the other ranks would idle at the barrier
        oname = "full" if args.aligner == "banded" else "banded"
        ocfg = default_config(aligner=oname, scoring_mode=args.mode, use_umi=args.umi, n_barcodes=args.barcodes, device=local_rank)
        octx = lib.Context(ocfg)
        octx.submit(batch)
        octx.run()
        t1 = time.perf_counter()
        for _ in range(max(1, min(args.steps, 3))):
            octx.run()
        dt = (time.perf_counter() - t1) / max(1, min(args.steps, 3))
        other = {"aligner": oname, "value": n_aln / dt, "unit": "read-alignments/s (1 GPU)", "ms_per_step": 1e3 * dt,
                 "sw_kernel_ms": octx.timing().sw_ms, "hard_tasks": octx.timing().hard_tasks}
        octx.close()

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = total_aln * args.steps / elapsed
        sw_avg_ms = float(np.mean(sw_ms))
        launches = ctx.timing().sw_launches
        alg_bytes = algorithmic_bytes(batch)
        achieved_gbs = alg_bytes / (sw_avg_ms * 1e-3) / 1e9
        traffic, lds_pmc = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                if j.get("workload_records") == batch.n_records:      # counters of exactly this workload (separate rocprofv3 --pmc runs)
                    traffic = j.get("hbm_bytes_per_launch")
                    lds_pmc = j.get("lds")
            except Exception:
                traffic, lds_pmc = None, None
        lane_ops = cells / 2 * OPS_PER_CELL_PAIR            # useful packed ops of rank 0's sw_full_kernel launch(es)
        full_avg_ms = float(np.mean(full_ms))
        out = {
            "metric": "read-alignments/sec at 100k loci x 10k cells; bit-exact .mtx vs ref",
            "value": value, "unit": "read-alignments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "i16x2 (packed int16 DP, int32 scores)", "data": "synthetic",
            "config": {"workload": spec.name + ", %s mode, %s aligner" % (args.mode, args.aligner),
                       "loci_per_gpu": args.loci, "barcodes": args.barcodes, "scored_reads_per_gpu": batch.n_records,
                       "alignments_per_step": total_aln, "dp_cells_per_step": total_cells, "triplets": total_nnz,
                       "sharding": "loci (matrix rows) per rank, COO rows gathered to rank 0" if world > 1 else "single GPU"},
            # dominant kernel: sw_full_duo_kernel (62 % of the step; profiles/r01_duo_kernel_stats_100k_banded.csv).  Its
            # duration is the ctx's hipEvent pair around the DP launches (vtx_timing.full_ms), live in this run.
            "roofline": {"bound": "hbm", "kernel": "sw_full_duo_kernel (1 launch per step)", "kernel_ms": full_avg_ms,
                         "achieved": alg_bytes / (full_avg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg_bytes / (full_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "lds_pmc": lds_pmc,
                         "whole_sw_stage": {"kernels": ("sw_full_duo_kernel" if args.aligner == "full" else "sw_full_duo_kernel + band_run_kernel + band_kernel + band_expand_kernel + sw_banded_kernel") + " (%d launches)" % launches,
                                            "ms": sw_avg_ms, "achieved": achieved_gbs},
                         "note": "integer DP: the binding roof is VALU issue, see roofline_valu; the HBM fraction is reported because north_star asks for it"},
            "roofline_valu": {"bound": "valu", "kernel": "sw_full_duo_kernel", "kernel_ms": full_avg_ms,
                              "note": "algorithmic ops = 9 packed ops x (read x haplotype cells of both alignments) / 2; the kernel "
                                      "shares the REF == ALT prefix columns between two reads, so it EXECUTES ~25 % fewer cell "
                                      "updates than that (DESIGN.md 4.1)",
                              "achieved": lane_ops / (full_avg_ms * 1e-3) / 1e12, "peak": VALU_PEAK_TOPS,
                              "unit": "T packed-lane-ops/s", "frac": lane_ops / (full_avg_ms * 1e-3) / 1e12 / VALU_PEAK_TOPS,
                              "gcups": cells / (full_avg_ms * 1e-3) / 1e9, "ops_per_cell_pair": OPS_PER_CELL_PAIR},
            "timing": {"sw_kernel_ms": sw_avg_ms, "full_kernel_ms": full_avg_ms, "band_kernels_ms": float(np.mean(band_ms)), "reduce_ms": float(np.mean(red_ms)), "submit_h2d_s": t_sub,
                       "generate_s": t_gen,
                       "pcie_inclusive_alignments_per_s": total_aln / world / (t_sub + elapsed / args.steps)},
        }
        if other is not None:
            out["other_aligner"] = other
        out["timing"]["hard_tasks"] = int(ctx.timing().hard_tasks)
        if not args.no_cpu_baseline and world == 1:                  # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(batch, cfg, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1 or force_gather:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
