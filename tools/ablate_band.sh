#!/bin/bash
# profiling aid: time band_fast_kernel with phases disabled (VTX_BAND_ABLATE: 1 tables only, 2 + probe loop, 3 full row loop without traceback/walk, 0 full)
for a in 1 2 3 0; do
  VTX_BAND_ABLATE=$a timeout 300 python bench.py --loci 10000 --barcodes 5000 --mode coverage --steps 3 --warmup 1 --no-cpu-baseline --no-other-aligner 2>/dev/null > /tmp/ab_$a.json
  python -c "import json;j=json.load(open('/tmp/ab_$a.json'));print('ablate $a band_kernels_ms', j['timing']['band_kernels_ms'])"
done
