set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r02m2
for MODE in 4; do
  for C in "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"; do
    N=$(echo $C | cut -d' ' -f1)
    timeout 200 rocprofv3 --kernel-trace --pmc $C -d gpurun_out/r02m2/p_${MODE}_$N -o pmc -- python tools/gemm_pmc_probe.py $MODE 3 > gpurun_out/r02m2/log_${MODE}_$N.txt 2>&1
    DB=$(find gpurun_out/r02m2/p_${MODE}_$N -name "*.db" | head -1)
    echo "== mode $MODE" >> gpurun_out/r02m2/summary.txt
    python tools/rocpd_counters.py $DB gemm >> gpurun_out/r02m2/summary.txt 2>&1
    python tools/rocpd_summary.py $DB | grep -i gemm | head -3 >> gpurun_out/r02m2/summary.txt 2>&1
  done
done
rm -rf gpurun_out/r02m2/p_*
cat gpurun_out/r02m2/summary.txt
