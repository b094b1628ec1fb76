#!/bin/bash
# usage (on the GPU box): bash tools/dbg/diag_pmc.sh OUTDIR — instruction / wait counters of band_diag_kernel stopped after the front (1),
# after the probes (2) and complete (0): what the probe phase costs in instructions as opposed to time
export TMPDIR=/tmp
OUT=$1; mkdir -p $OUT
for a in 1 2 0; do
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD"; do
    g=$(echo $grp | cut -d' ' -f1)
    VTX_DIAG_ABLATE=$a rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_${a}_$g -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-aligner --no-sensitivity > /dev/null 2> $OUT/pmc_${a}_$g.err
    echo "== ablate $a" >> $OUT/pmc.txt
    python tools/pmc_quick.py /tmp/pmc_${a}_$g band_diag >> $OUT/pmc.txt
  done
done
cat $OUT/pmc.txt
