# phases of band_coop_kernel on the real-sequence workload (VTX_COOP_ABLATE: 1 = matches only, 2 = + sdpkpp, 0 = all; results of 1 / 2 are wrong by design)
cd /tmp && export TMPDIR=/tmp
for a in ${ABLATES:-1 2 0}; do
  rm -rf /tmp/profc
  VTX_COOP_ABLATE=$a rocprofv3 --kernel-trace --stats -d /tmp/profc -o c --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --genome $GRAFT_REPO_ROOT/tests/golden/test_dna.fa --loci 30000 --steps 2 --warmup 1 --no-cpu-baseline --no-other-aligner > /dev/null 2>&1
  python -c "
import csv,glob
f=glob.glob('/tmp/profc/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f))):
    if 'coop' in r['Name'] or 'sw_banded' in r['Name'] or 'band_run' in r['Name']: print('ablate $a', r['Name'][:44], r['Calls'], round(float(r['TotalDurationNs'])/1e6/3,2), 'ms/step')
"
done
