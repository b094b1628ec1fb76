import os, sys, time
sys.path.insert(0, os.getcwd())
from oracle import oracle
from vartrix_amd import synth
from vartrix_amd.abi import default_config
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
spec = synth.SynthSpec(n_loci=2048, n_barcodes=1000, reads_per_locus=256, seed=1)
b = synth.make_batch(spec)
cfg = default_config(aligner="banded", n_barcodes=1000)
for th in (1, 8, 32, 64, 128, 256):
    n = min(2048, max(8, th * 8))
    s = b.slice_loci(0, n)
    t0 = time.perf_counter(); oracle.batch_scores(s, cfg, threads=th); dt = time.perf_counter() - t0
    print(th, "threads:", n, "loci", "%.2f s" % dt, "%.0f aln/s" % (2 * s.n_records / dt), "per thread %.0f" % (2 * s.n_records / dt / th))
