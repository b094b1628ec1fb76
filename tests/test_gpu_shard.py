"""The RCCL leg on the GPU box (one rank — the box has one GPU): `nccl` process group, the context's device-resident
triplets wrapped as torch tensors through their raw device pointers, shard.GatherPipeline — compared bit for bit with
vtx_fetch_coo; and bench.py's own gather path (VTX_FORCE_GATHER=1) against its plain single-GPU run.  The N > 1
exchange itself is covered by the world_size-2 gloo tests (tests/test_shard.py), which drive the same objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CODE = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from vartrix_amd import lib, shard, synth
from vartrix_amd.abi import MODES, default_config
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29631"
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for mode, umi in (("consensus", 0), ("alt_frac", 1), ("coverage", 1)):
    spec = synth.SynthSpec(n_loci=300, n_barcodes=500, reads_per_locus=40, use_umi=bool(umi), indel_frac=0.3, seed=9)
    batch = synth.make_batch(spec)
    cfg = default_config(aligner="banded", scoring_mode=mode, use_umi=umi, n_barcodes=500)
    with lib.Context(cfg) as ctx:
        ctx.submit(batch)
        pipe = shard.GatherPipeline(MODES[mode])
        for _ in range(3):
            ctx.run()
            pipe.push(shard.device_coo_tensors(ctx, dev))
        got = shard.tensors_to_coo(pipe.drain())
        want = ctx.fetch_coo()
    assert pipe.completed == 3
    for k in want:
        assert np.array_equal(got[k].view(np.uint8), want[k].view(np.uint8)), (mode, k)
    assert len(want["row"]) > 1000
dist.barrier(); dist.destroy_process_group()
print("rccl-one-rank-ok")
''' % ROOT


def test_rccl_gather_one_rank_equals_fetch_coo():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _CODE], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "rccl-one-rank-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def _bench(extra_env, args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT="29633", **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-other-aligner"] + args, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads(lines[-1])


def test_bench_gather_path_matches_plain_run():
    """bench.py with the RCCL gather forced on one rank must report the same matrix summary (nnz, checksum) as the
    plain run of the same workload, in config-4 shape (many barcodes) at reduced size."""
    args = ["--loci", "1500", "--barcodes", "5000", "--reads-per-locus", "64"]
    plain = _bench({}, args)
    gath = _bench({"VTX_FORCE_GATHER": "1"}, args)
    assert plain["result"] == gath["result"] and plain["result"]["nnz"] > 10000
    assert plain["config"]["alignments_per_step"] == gath["config"]["alignments_per_step"]
    for j in (plain, gath):
        assert j["roofline"]["kernel"].startswith(("band_diag_kernel", "band_run_kernel")) and j["roofline"]["kernel_ms"] > 0
        assert "roofline_valu" not in j or j["roofline_valu"]["frac"] <= 1.0      # no fraction above 1 under a key called frac
        assert j["scaling"] == "strong" and j["n_gpus"] == 1


_NATIVE = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
from vartrix_amd import lib, synth
from vartrix_amd.abi import default_config
ident = lib.comm_id()
assert len(ident) == 128
for mode, umi in (("consensus", 0), ("alt_frac", 1), ("coverage", 1)):
    spec = synth.SynthSpec(n_loci=300, n_barcodes=500, reads_per_locus=40, use_umi=bool(umi), indel_frac=0.3, seed=9)
    batch = synth.make_batch(spec)
    cfg = default_config(aligner="banded", scoring_mode=mode, use_umi=umi, n_barcodes=500)
    with lib.Context(cfg) as ctx:
        ctx.comm_init(ident if mode == "consensus" else lib.comm_id(), 0, 1)
        ctx.submit(batch)
        for _ in range(2):
            ctx.run()
            d = ctx.gather_coo(0)
        got = ctx.fetch_gathered()
        want = ctx.fetch_coo()
    assert d["nnz"] == len(want["row"]) > 1000
    for k in want:
        assert np.array_equal(got[k].view(np.uint8), want[k].view(np.uint8)), (mode, k)
# status round: a rank without a completed vtx_run reports it instead of entering the exchange (the other ranks would get
# VTX_E_PEER); vtx_gather_abort is the same round for a rank whose own work failed
from vartrix_amd import abi
with lib.Context(default_config(aligner="banded", n_barcodes=500)) as ctx:
    ctx.comm_init(lib.comm_id(), 0, 1)
    ctx.submit(batch)
    try:
        ctx.gather_coo(0)
        raise SystemExit("gather_coo before vtx_run must fail")
    except lib.VtxError as e:
        assert e.status == abi.VTX_E_STATE, e
    ctx.gather_abort()
    ctx.run()
    assert ctx.gather_coo(0)["nnz"] > 1000
print("native-gather-one-rank-ok")
''' % ROOT


def test_native_rccl_gather_one_rank_equals_fetch_coo():
    """The exchange behind the C-ABI (vtx_comm_init / vtx_gather_coo: RCCL all-gather of the counts, point-to-point blocks,
    values recomputed from the counts on the destination) with a one-rank communicator: bit for bit vtx_fetch_coo."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _NATIVE], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "native-gather-one-rank-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_torch_gather_matches_plain_run():
    """The alternative exchange in Python (VTX_TORCH_GATHER=1: shard.GatherPipeline over torch.distributed) gives the same matrix;
    the default — asserted here too — is the library's own vtx_gather_coo."""
    args = ["--loci", "1500", "--barcodes", "5000", "--reads-per-locus", "64"]
    plain = _bench({}, args)
    nat = _bench({"VTX_FORCE_GATHER": "1"}, args)
    alt = _bench({"VTX_FORCE_GATHER": "1", "VTX_TORCH_GATHER": "1"}, args)
    assert plain["result"] == nat["result"] == alt["result"] and nat["result"]["nnz"] > 10000
    assert plain["gather"] is None
    assert nat["gather"]["impl"].startswith("vtx_gather_coo") and nat["gather"]["transport"] == "RCCL" and nat["gather"]["ranks"] == 1
    assert alt["gather"]["impl"].startswith("torch.distributed") and alt["gather"]["fallback_from_vtx_gather_coo"] is None
    # a communicator that cannot be set up: every rank falls back to the torch.distributed gather together, and the line says so
    fb = _bench({"VTX_FORCE_GATHER": "1", "VTX_BENCH_TEST_COMM_FAIL": "1"}, args)
    assert fb["result"] == plain["result"] and fb["gather"]["impl"].startswith("torch.distributed")
    assert "VTX_BENCH_TEST_COMM_FAIL" in fb["gather"]["fallback_from_vtx_gather_coo"]


def test_two_rank_launch_rehearsal_on_one_device(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` exactly as the driver launches the scaling bench,
    on ONE device: the ranks share it, the library's exchange (vtx_comm_init, vtx_gather_coo in every step) runs over the test
    transport of libvtx_dev.so, torch.distributed falls back to gloo for the barriers.  The line must parse, say n_gpus 2 and the
    library's gather, and carry the matrix of the unsharded run."""
    args = ["--loci", "4000", "--barcodes", "50000", "--reads-per-locus", "32"]
    plain = _bench({}, args)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VTX_LIB_VARIANT="dev", VTX_COMM_TEST_TRANSPORT=str(tmp_path))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29641", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"] + args,
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["scaling"] == "strong"
    assert j["gather"]["impl"].startswith("vtx_gather_coo") and j["gather"]["ranks"] == 2 and j["gather"]["rccl_ranks"] == 2
    assert j["result"] == plain["result"] and j["result"]["nnz"] > 50000
    assert j["config"]["alignments_per_step"] == plain["config"]["alignments_per_step"]
    assert j["value"] > 0 and j["ms_per_step"] > 0 and j["sustained"]["steps"] >= 1


def test_eight_rank_launch_rehearsal_on_one_device(tmp_path):
    """The driver's 8-GPU command — `torchrun --nproc-per-node 8 bench.py --gpus 8` — on ONE device over the test transport (round 6:
    8 is what the scaling run launches; 2 and 4 were the rehearsed worlds).  Eight contexts share the device (a reduced config 4:
    50 000 barcodes, sharded by partition_loci), every step ends in the library's gather with seven senders, the communicator
    reports 8 ranks, and the gathered matrix equals the unsharded run's."""
    args = ["--loci", "6000", "--barcodes", "50000", "--reads-per-locus", "24", "--no-sensitivity", "--no-cpu-baseline", "--no-other-aligner"]
    plain = _bench({}, args)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VTX_LIB_VARIANT="dev", VTX_COMM_TEST_TRANSPORT=str(tmp_path))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", "29647", os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                        "--sustain-seconds", "0"] + args,
                       capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["steps"] == 2 and j["scaling"] == "strong"
    assert j["gather"]["impl"].startswith("vtx_gather_coo") and j["gather"]["ranks"] == 8 and j["gather"]["rccl_ranks"] == 8
    assert j["result"] == plain["result"] and j["result"]["nnz"] > 50000
    assert j["config"]["alignments_per_step"] == plain["config"]["alignments_per_step"]


# ---- vtx_gather_coo with world 2 and 4: ranks = processes on ONE device, RCCL's nine entry points replaced by the test transport
#      (vartrix_amd/csrc/vtx_comm_test.hip, VTX_COMM_TEST_TRANSPORT).  What runs is the library's exchange as shipped: the status
#      rounds, vtx_gather_plan, the grouped Send / Recv of the five arrays to their final offsets on rank 0, the f64 values
#      recomputed there — with a real second (third, fourth) rank on the other side of every Recv.  RCCL itself and xGMI do not. ----
_RANK_CODE = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
from vartrix_amd import lib, shard, synth
from vartrix_amd.abi import default_config
rank, world, scenario, tmp = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
mode, umi = ("alt_frac", 1) if scenario != "consensus" else ("consensus", 0)
spec = synth.SynthSpec(n_loci=240, n_barcodes=300, reads_per_locus=24, use_umi=bool(umi), indel_frac=0.3, seed=17)
whole = synth.make_batch(spec)
parts = shard.partition_loci(whole, world)
if scenario == "empty":                      # rank 1 holds no locus at all
    parts = [(0, parts[1][1])] + [(parts[1][1], parts[1][1])] + parts[2:]
lo, hi = parts[rank]
mine = whole.slice_loci(lo, hi)
idf = os.path.join(tmp, "id")
if rank == 0:
    ident = lib.comm_id()
    open(idf + ".tmp", "wb").write(ident); os.rename(idf + ".tmp", idf)
else:
    t0 = time.time()
    while not os.path.exists(idf):
        assert time.time() - t0 < 60
        time.sleep(0.01)
    ident = open(idf, "rb").read()
cfg = default_config(aligner="banded", scoring_mode=mode, use_umi=umi, n_barcodes=300)
with lib.Context(cfg) as ctx:
    ctx.comm_init(ident, rank, world)
    ctx.submit(mine)
    if scenario == "fail" and rank == world - 1:
        ctx.gather_abort()                   # "my own work failed": the others must leave gather_coo with VTX_E_PEER
        print("aborted"); sys.exit(0)
    ctx.run()
    try:
        g = ctx.gather_coo(0)
    except lib.VtxError as e:
        print("error", e.status); sys.exit(0)
    if rank == 0:
        got = ctx.fetch_gathered()
        np.savez(os.path.join(tmp, "gathered.npz"), **got)
    print("ok", g["nnz"])
''' % ROOT


def _run_world(world, scenario, tmp):
    env = dict(os.environ, VTX_LIB_VARIANT="dev", VTX_COMM_TEST_TRANSPORT=str(tmp))   # (the test transport is in libvtx_dev.so only)
    procs = [subprocess.Popen([sys.executable, "-c", _RANK_CODE, str(r), str(world), scenario, str(tmp)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, e[-3000:]
        outs.append(o.strip().splitlines()[-1])
    return outs


@pytest.mark.parametrize("world,scenario", [(2, "consensus"), (2, "alt_frac"), (4, "alt_frac"), (4, "empty")])
def test_gather_coo_between_processes_equals_the_unsharded_matrix(tmp_path, world, scenario):
    import numpy as np
    from vartrix_amd import lib, synth
    from vartrix_amd.abi import default_config
    outs = _run_world(world, scenario, tmp_path)
    assert all(o.startswith("ok") for o in outs), outs
    assert [int(o.split()[1]) for o in outs[1:]] == [0] * (world - 1)                  # nnz = 0 away from the destination
    mode, umi = ("alt_frac", 1) if scenario != "consensus" else ("consensus", 0)
    spec = synth.SynthSpec(n_loci=240, n_barcodes=300, reads_per_locus=24, use_umi=bool(umi), indel_frac=0.3, seed=17)
    with lib.Context(default_config(aligner="banded", scoring_mode=mode, use_umi=umi, n_barcodes=300)) as ctx:
        ctx.submit(synth.make_batch(spec))
        ctx.run()
        want = ctx.fetch_coo()
    got = np.load(tmp_path / "gathered.npz")
    assert int(outs[0].split()[1]) == len(want["row"]) > 1000
    for k in want:
        assert np.array_equal(got[k].view(np.uint8), want[k].view(np.uint8)), (world, scenario, k)


def test_a_failing_rank_makes_every_gather_return_peer_error(tmp_path):
    from vartrix_amd import abi
    outs = _run_world(4, "fail", tmp_path)
    assert outs[3] == "aborted"
    assert outs[:3] == ["error %d" % abi.VTX_E_PEER] * 3, outs
