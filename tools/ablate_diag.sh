for a in ${ABLATES:-1 2 0}; do
  VTX_DIAG_ABLATE=$a timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-aligner 2>/dev/null > /tmp/ab_$a.json
  python -c "import json;j=json.load(open('/tmp/ab_$a.json'));print('diag ablate $a', j['timing']['band_diag_ms'], j['ms_per_step'])"
done
