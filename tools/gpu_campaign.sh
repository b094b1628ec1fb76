#!/bin/bash
# One script for the measurement campaigns on the GPU box (run from the repo root, e.g. through gpurun).  Everything lands in
# OUTDIR (use gpurun_out/<name>); copy what is to be judged into profiles/.
#
#   bash tools/gpu_campaign.sh stats   OUTDIR WORKLOAD...   rocprofv3 --kernel-trace --stats of bench.py (3 steps), one CSV + bench line per
#                                                           workload; prints the kernels above 0.3 % of the run
#   bash tools/gpu_campaign.sh lines   OUTDIR WORKLOAD...   bench lines only (no profiler), appended to OUTDIR/lines.jsonl
#   bash tools/gpu_campaign.sh ablate  OUTDIR KNOB VALUES.. [-- bench args]   a kernel's time when it stops after each phase: KNOB is one of
#                                                           VTX_DIAG_ABLATE / VTX_SWEEP_ABLATE / VTX_BAND_ABLATE / VTX_COOP_ABLATE — these exist
#                                                           in libvtx_dev.so only (the script sets VTX_LIB_VARIANT=dev); scores are wrong by
#                                                           design for values != 0, only the kernel times are read
#   bash tools/gpu_campaign.sh variants OUTDIR WORKLOAD LIB...   the same workload on several builds of the library, one line each: LIB is
#                                                           prod (libvtx.so) or a VTX_LIB_VARIANT name (dev, lazy0, ..., or an experimental
#                                                           libvtx_<name>.so built by hand: how S2_WORDS / the stream kernel's occupancy were chosen)
#   bash tools/gpu_campaign.sh final   OUTDIR               the round's closing set: default bench line, stats of every workload, the depth
#                                                           ladder, config 4 on one GPU, the PMC passes (tools/pmc_collect.sh), the CLI end to end
#
# WORKLOAD names: head (config 3, the headline) | genome (loci from tests/golden/test_dna.fa) | e1 e3 e8 (1 / 3 / 8 % substitution errors) |
#   c5 (config-5 shape: 30 % indel loci, UMIs, alt_frac) | d128 d64 d32 d16 d4 (reads per locus) | ln8 (log-normal depth, median 8) |
#   r250 (250-base reads) | p150 (--padding 150: haplotypes of 301 bases) | c4 (config 4 on one GPU)
set -u
export TMPDIR=/tmp
CMD=$1; OUT=$2; shift 2
mkdir -p "$OUT"
BENCH="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-aligner --no-sensitivity --sustain-seconds 0"

wl_args() {
  case $1 in
    head) echo "";; genome) echo "--genome tests/golden/test_dna.fa";;
    e1) echo "--sub-error 0.01 --loci 100000";; e3) echo "--sub-error 0.03 --loci 100000";; e8) echo "--sub-error 0.08 --loci 100000";;
    c5) echo "--indel-frac 0.3 --umi 1 --mode alt_frac --loci 100000";;
    d128|d64|d32|d16|d4) echo "--reads-per-locus ${1#d}";; ln8) echo "--reads-per-locus 8 --depth-sigma 1.0";;
    r250) echo "--read-len 250 --loci 60000";; p150) echo "--padding 150 --loci 100000";; c4) echo "--workload config4";;
    *) echo "unknown workload $1" >&2; exit 2;;
  esac
}
kstats() {   # file.csv runs
  python - "$1" "$2" <<'PY'
import csv, sys
runs = int(sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['Percentage']) < 0.3: continue
    ms = float(r['AverageNs']) / 1e6
    print("  %-58s calls/run %6.1f avg %9.3f ms  per-run %9.2f ms  %5.1f %%" % (r['Name'].split('(')[0][-58:], int(r['Calls']) / runs, ms, ms * int(r['Calls']) / runs, float(r['Percentage'])))
PY
}
show_line() {   # bench json file
  python - "$1" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    t = j['timing']
    print('  %.2f ms/step  %.3e aln/s' % (j['ms_per_step'], j['value']), {k: t[k] for k in ('diag_left_tasks', 'checked_tasks', 'swept_tasks', 'hard_tasks', 'overflow_tasks')}, j['result'])
except Exception as e:
    print('  no bench line:', e)
PY
}

case $CMD in
  stats)
    for w in "$@"; do
      A=$(wl_args $w) || exit 2
      rm -rf /tmp/prof_$w
      rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o $w -- $BENCH $A > $OUT/$w.json 2> $OUT/$w.err; echo "== $w rc=$?"
      find /tmp/prof_$w -name "*kernel_stats.csv" -exec cp {} $OUT/${w}_kernel_stats.csv \;
      show_line $OUT/$w.json; [ -f $OUT/${w}_kernel_stats.csv ] && kstats $OUT/${w}_kernel_stats.csv 4 | head -${N:-10}
    done;;
  lines)
    for w in "$@"; do
      A=$(wl_args $w) || exit 2
      python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-aligner --no-sensitivity $A 2> $OUT/$w.err | tail -1 > $OUT/$w.json
      echo "== $w"; show_line $OUT/$w.json; cat $OUT/$w.json >> $OUT/lines.jsonl
    done;;
  ablate)
    KNOB=$1; shift; VALS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done; [ $# -gt 0 ] && shift
    export VTX_LIB_VARIANT=dev
    for a in "${VALS[@]}"; do
      rm -rf /tmp/abl_$a
      env $KNOB=$a rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_$a -o a -- $BENCH "$@" > $OUT/abl_$a.json 2> $OUT/abl_$a.err
      f=$(find /tmp/abl_$a -name "*kernel_stats.csv" | head -1); echo "== $KNOB=$a" | tee -a $OUT/ablate.txt; kstats "$f" 4 | head -6 | tee -a $OUT/ablate.txt
    done;;
  variants)
    W=$1; shift
    A=$(wl_args $W) || exit 2
    for v in "$@"; do
      if [ "$v" = prod ]; then unset VTX_LIB_VARIANT; else export VTX_LIB_VARIANT=$v; fi
      $BENCH $A 2> $OUT/var_$v.err | tail -1 > $OUT/var_$v.json
      echo "== $W on $v"; show_line $OUT/var_$v.json
    done;;
  final)
    # the counters first: bench.py's line carries roofline.traffic only from a profiles/pmc_traffic.json of the sources it runs
    bash tools/pmc_collect.sh /tmp/pmc --no-sensitivity --sustain-seconds 0 > $OUT/pmc_collect.log 2>&1
    python tools/pmc_summarize.py /tmp/pmc $OUT/pmc_counters.json 24320920 $OUT/pmc_traffic.json > $OUT/pmc_summary.txt 2>&1
    [ -s $OUT/pmc_traffic.json ] && cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
    python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?"
    bash $0 stats $OUT head genome e3 e8 c5 d16 d4 e1 r250 p150
    bash $0 lines $OUT d128 d64 d32 d16 d4 ln8 c4
    # end to end through the drop-in CLI at config-3 scale (device ingest against the host packer), and the kernels of one such run
    VTXH_PROFILE=1 python tools/e2e_cli_bench.py --fast --loci 100000 --reads 256 --barcodes 10000 --out /tmp/e2e > $OUT/e2e_cli_config3.log 2>&1; echo "e2e rc=$?"
    cp /tmp/e2e/e2e_summary.json $OUT/e2e_summary.json 2>/dev/null
    bash tools/e2e_profile.sh $OUT/e2e_prof > $OUT/e2e_profile.txt 2>&1; tail -14 $OUT/e2e_profile.txt
    echo done;;
  *) echo "unknown command $CMD" >&2; exit 2;;
esac
