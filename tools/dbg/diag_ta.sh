#!/bin/bash
# usage (on the GPU box): bash tools/dbg/diag_ta.sh OUTDIR — address-unit / L1 counters of band_diag_kernel, complete and stopped after the front
export TMPDIR=/tmp
OUT=$1; mkdir -p $OUT
for a in 0 1; do
  for grp in "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TA_TCP_STATE_READ_sum"; do
    g=$(echo $grp | cut -d' ' -f1)
    VTX_DIAG_ABLATE=$a rocprofv3 --pmc $grp --output-format csv -d /tmp/ta_${a}_$g -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-aligner --no-sensitivity > /dev/null 2> $OUT/ta_${a}_$g.err
    echo "== ablate $a" >> $OUT/ta.txt
    python tools/pmc_quick.py /tmp/ta_${a}_$g band_diag >> $OUT/ta.txt
  done
done
cat $OUT/ta.txt
