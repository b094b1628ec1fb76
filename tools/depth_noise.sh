mkdir -p gpurun_out/r3s
: > gpurun_out/r3s/depth_noise.jsonl
for args in "--reads-per-locus 4" "--reads-per-locus 16" "--reads-per-locus 32" "--reads-per-locus 64" "--reads-per-locus 128" "--reads-per-locus 8 --depth-sigma 1.0" "--sub-error 0.01" "--sub-error 0.03" "--sub-error 0.08" "--indel-frac 0.3 --umi 1 --mode alt_frac"; do
  timeout 300 python bench.py --loci 100000 $args --steps 5 --warmup 2 --no-cpu-baseline --no-other-aligner 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(json.dumps({'args':'$args','value':j['value'],'ms_per_step':j['ms_per_step'],'timing':j['timing'],'alignments':j['config']['alignments_per_step']}))" >> gpurun_out/r3s/depth_noise.jsonl
done
cat gpurun_out/r3s/depth_noise.jsonl | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); t=j['timing']; print(j['args'], '%.3g aln/s'%j['value'], '%.1f ms'%j['ms_per_step'], 'diag %.1f run %.1f left %d hard %d over %d'%(t['band_diag_ms'],t['band_run_kernel_ms'],t['diag_left_tasks'],t['hard_tasks'],t['overflow_tasks']), j['alignments'])"
