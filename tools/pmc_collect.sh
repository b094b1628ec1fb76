#!/bin/bash
# PMC counters of one vtx_run of the bench workload, one counter group per rocprofv3 pass
# (--pmc is never combined with sys/hip/hsa tracing; FETCH_SIZE and WRITE_SIZE need 3 + 2 of the 4 TCC counters, so
# they cannot share a pass).  PASSES="1 2" restricts the run to some passes.  Run on the GPU box from the repo root:
#     bash tools/pmc_collect.sh gpurun_out/pmc [bench.py args...]
# then: python tools/pmc_summarize.py gpurun_out/pmc profiles/<name>.json
# Passes 6-8 (round 3): the VALU busy cycles behind roofline_valu_issue — SQ_ACTIVE_INST_VALU (quad-cycles a wave spends
# executing VALU instructions, summed over waves: at most one wave per SIMD executes one at a time, so x 4 / (SIMDs x
# kernel cycles) is the VALU pipe occupancy), SQ_THREAD_CYCLES_VALU (the same weighted by active lanes), the SQ busy
# cycles and the GRBM clock count the kernel's effective clock comes from, and the raw TCC request counters FETCH_SIZE is
# derived from.
set -u
OUT=$(realpath -m "$1"); shift
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES" \
           "SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum"; do
    i=$((i + 1))
    if [ -n "${PASSES:-}" ] && ! echo " $PASSES " | grep -q " $i "; then continue; fi
    timeout 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pass$i" -- \
        python "$REPO/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-other-aligner "$@" > "$OUT/pass$i.log" 2>&1
    echo "pass $i ($grp): rc=$?"
done
