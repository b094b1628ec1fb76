for l in ${LANES:-64 16 4 1}; do
  VTX_BAND_LANES=$l timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-aligner 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); t=j['timing']; print('band_lanes $l', '%.2f ms'%j['ms_per_step'], 'kernels %.2f diag %.2f run %.2f over %d'%(t['band_kernels_ms'], t['band_diag_ms'],t['band_run_kernel_ms'],t['overflow_tasks']), j['result'])"
done
