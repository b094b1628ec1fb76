for args in "--sub-error 0.005" "--sub-error 0.01" "--sub-error 0.03" "--sub-error 0.08"; do
  timeout 300 python bench.py --loci 100000 $args --steps 3 --warmup 1 --no-cpu-baseline --no-other-aligner 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); t=j['timing']; print('$args', '%.3g aln/s'%j['value'], '%.1f ms'%j['ms_per_step'], 'diag %.1f run %.1f left %d hard %d over %d'%(t['band_diag_ms'],t['band_run_kernel_ms'],t['diag_left_tasks'],t['hard_tasks'],t['overflow_tasks']))"
done
