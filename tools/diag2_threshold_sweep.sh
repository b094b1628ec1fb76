export VTX_LIB_VARIANT=dev
for n in 500 1000 2000 5000 10000 20000; do
  for m in 1 2000000000; do
    VTX_BAND_DIAG2_MIN=$m python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-aligner --no-sensitivity --sustain-seconds 0 --genome tests/golden/test_dna.fa --loci $n 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=j['timing']
print('loci $n diag2_min $m: %.3f ms/step  second_stage %d swept %d left %d' % (j['ms_per_step'], t['second_stage_tasks'], t['swept_tasks'], t['diag_left_tasks']))"
  done
done
