export TMPDIR=/tmp VTX_LIB_VARIANT=dev
cd /tmp
for a in 2 3 0; do
  for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_SALU" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
    g=$(echo $grp | cut -d' ' -f1)
    rm -rf /tmp/pmc_${a}_$g
    VTX_SWEEP_ABLATE=$a rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_${a}_$g -o p -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-aligner --no-sensitivity --sustain-seconds 0 --genome /root/repo/tests/golden/test_dna.fa --loci 30000 > /dev/null 2>&1
    echo "== ablate $a"; python /root/repo/tools/pmc_quick.py /tmp/pmc_${a}_$g band_sweep
  done
done
