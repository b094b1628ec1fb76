#!/bin/bash
# usage (on the GPU box): bash tools/dbg/diag_ablate.sh OUTDIR [bench args] — band_diag_kernel's time when it stops after each phase
# (VTX_DIAG_ABLATE: 4 set-up, 3 + diagonal and mask, 1 + front, 2 + probes, 7 + sort, 5 + harmless tests, 6 + closure, 0 everything;
# the scores are wrong by design, only the kernel's time is read)
export TMPDIR=/tmp
OUT=$1; shift; mkdir -p $OUT
for a in ${ABL:-4 3 1 2 7 5 6 0}; do
  VTX_DIAG_ABLATE=$a rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_$a -o a -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-aligner --no-sensitivity "$@" > $OUT/abl_$a.json 2> $OUT/abl_$a.err
  f=$(find /tmp/abl_$a -name "*kernel_stats.csv" | head -1)
  python - "$f" $a <<'PY' | tee -a $OUT/ablate.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'band_diag_kernel' in r['Name']:
        print('ablate %s: band_diag_kernel %.3f ms (%s calls)' % (sys.argv[2], float(r['AverageNs']) / 1e6, r['Calls']))
PY
done
