#!/bin/bash
# usage (on the GPU box): bash tools/dbg/run4.sh OUTDIR [workloads...]  — kernel stats of the four reference workloads
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
for w in "$@"; do
  case $w in head) A="";; genome) A="--genome tests/golden/test_dna.fa";; e3) A="--sub-error 0.03 --loci 100000";; e8) A="--sub-error 0.08 --loci 100000";; e1) A="--sub-error 0.01 --loci 100000";; d16) A="--reads-per-locus 16";; d4) A="--reads-per-locus 4";; c5) A="--indel-frac 0.3 --umi 1 --mode alt_frac --loci 100000";; esac
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o $w -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-aligner --no-sensitivity $A > $OUT/$w.json 2> $OUT/$w.err; echo "$w rc=$?"
  find /tmp/prof_$w -name "*kernel_stats.csv" -exec cp {} $OUT/${w}_kernel_stats.csv \;
done
