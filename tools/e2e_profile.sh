#!/bin/bash
# Kernel-level profile of ONE drop-in CLI run at config-3 scale (device ingest): authors the inputs with tools/e2e_cli_bench.py, then
# runs bin/vartrix under VTXH_PROFILE=1 rocprofv3 --kernel-trace --stats.  GPU box, from the repo root:   bash tools/e2e_profile.sh gpurun_out/e2e_prof
set -u
OUT=$(realpath -m "$1"); mkdir -p "$OUT"
REPO=$(pwd)
export TMPDIR=/tmp
[ -f /tmp/e2e/r.bam ] || python tools/e2e_cli_bench.py --fast --loci 100000 --reads 256 --barcodes 10000 --out /tmp/e2e > "$OUT/e2e.log" 2>&1 || { tail -5 "$OUT/e2e.log"; exit 1; }
cd /tmp
rm -rf /tmp/e2e_prof /tmp/e2e/p.mtx /tmp/e2e/ref_matrix.mtx
VTXH_PROFILE=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/e2e_prof -o cli -- "$REPO/vartrix_amd/bin/vartrix" -v /tmp/e2e/v.vcf -b /tmp/e2e/r.bam \
    -f /tmp/e2e/g.fa -c /tmp/e2e/bcs.tsv -o /tmp/e2e/p.mtx --threads 16 --log-level info > "$OUT/cli.log" 2>&1
echo "rc=$?"
find /tmp/e2e_prof -name "*kernel_stats.csv" -exec cp {} "$OUT/cli_kernel_stats.csv" \;
python - "$OUT/cli_kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['Percentage']) < 0.5: continue
    print("  %-60s calls %4s avg %9.3f ms total %9.3f ms %5.1f %%" % (r['Name'].split('(')[0][-60:], r['Calls'], float(r['AverageNs']) / 1e6, float(r['TotalDurationNs']) / 1e6, float(r['Percentage'])))
PY
