#!/bin/bash
# profiling aid: time band_run_kernel with phases disabled (VTX_BAND_ABLATE: 1 tables only, 2 + probe loop,
# 3 phase 1 complete, 4 + phase 2 and the run bound, no staircase walk, 0 everything).  Scores are meaningless for != 0.
# usage: bash tools/ablate_band.sh [bench.py args]   (default: config 3)
for a in 1 2 3 4 0; do
  VTX_BAND_ABLATE=$a timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-aligner "$@" 2>/dev/null > /tmp/ab_$a.json
  python -c "import json;j=json.load(open('/tmp/ab_$a.json'));print('ablate $a band_kernels_ms', j['timing']['band_kernels_ms'], 'hard', j['timing']['hard_tasks'])"
done
