#!/bin/bash
# usage (on the GPU box): bash tools/dbg/depth.sh OUTDIR — the depth ladder without profiler (bench lines), then kernel stats at 16 reads per locus
export TMPDIR=/tmp
OUT=$1; mkdir -p $OUT
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-aligner --no-sensitivity"
$B --reads-per-locus 4 > $OUT/d4.json 2> $OUT/d4.err; echo "d4 rc=$?"
$B --reads-per-locus 16 > $OUT/d16.json 2> $OUT/d16.err; echo "d16 rc=$?"
$B --reads-per-locus 8 --depth-sigma 1.0 > $OUT/ln8.json 2> $OUT/ln8.err; echo "ln8 rc=$?"
$B > $OUT/head.json 2> $OUT/head.err; echo "head rc=$?"
VTX_BAND_TABLES_V1=1 $B --reads-per-locus 16 > $OUT/d16_v1.json 2> $OUT/d16_v1.err; echo "d16 v1 rc=$?"
bash tools/dbg/run4.sh $OUT d16 d4
