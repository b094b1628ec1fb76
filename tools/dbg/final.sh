#!/bin/bash
# final measurement campaign of a round (GPU box, repo root): bash tools/dbg/final.sh gpurun_out/final
export TMPDIR=/tmp
OUT=$1; mkdir -p $OUT
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?"
bash tools/dbg/run4.sh $OUT head genome e3 e8 c5 d16 d4 e1
for a in "--reads-per-locus 128" "--reads-per-locus 64" "--reads-per-locus 32" "--reads-per-locus 16" "--reads-per-locus 4" "--reads-per-locus 8 --depth-sigma 1.0"; do
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-aligner --no-sensitivity $a 2>/dev/null | tail -1 >> $OUT/depth.jsonl
done
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-aligner --no-sensitivity --workload config4 2>/dev/null | tail -1 > $OUT/config4_1gpu.json
bash tools/pmc_collect.sh /tmp/pmc --no-sensitivity > $OUT/pmc_collect.log 2>&1
python tools/pmc_summarize.py /tmp/pmc $OUT/pmc_counters.json 24320920 $OUT/pmc_traffic.json > $OUT/pmc_summary.txt 2>&1
echo done
