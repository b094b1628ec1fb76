#!/bin/bash
# usage: abvar.sh variant...   (real-sequence workload, 3 steps each)
for v in "$@"; do
  if [ "$v" = prod ]; then unset VTX_LIB_VARIANT; else export VTX_LIB_VARIANT=$v; fi
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-aligner --genome tests/golden/test_dna.fa --no-sensitivity --sustain-seconds 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); t=d['timing']; print('$v', round(d['ms_per_step'],2), 'ms', d['result'], {k:t[k] for k in ('swept_tasks','checked_tasks') if k in t})
"
done
