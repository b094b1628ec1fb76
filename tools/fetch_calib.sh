#!/bin/bash
# FETCH_SIZE / TCC request counters on access shapes of known size (tools/fetch_calib.hip).  On the GPU box:
#     bash tools/fetch_calib.sh gpurun_out/calib        then: python tools/pmc_summarize.py --calib gpurun_out/calib profiles/<name>.json
set -u
OUT=$(realpath -m "$1")
REPO=$(pwd)
mkdir -p "$OUT"
BIN="$REPO/tools/bin/fetch_calib"
[ -x "$BIN" ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o "$BIN" "$REPO/tools/fetch_calib.hip" || exit 1
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i + 1))
    timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pass$i" -- "$BIN" > "$OUT/pass$i.log" 2>&1
    echo "calib pass $i ($grp): rc=$?"
done
grep -h CALIB "$OUT/pass1.log" > "$OUT/known_bytes.txt"
