set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r02m16b; mkdir -p $O; rm -f $O/pmc_summary.txt
for C in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/pmc_$N -o pmc -- python tools/attn_pmc_probe.py prescaled 3 > $O/pmc_$N.log 2>&1
  DB=$(find $O/pmc_$N -name "*.db" | head -1); python tools/rocpd_counters.py $DB attn4 >> $O/pmc_summary.txt 2>&1
  python tools/rocpd_summary.py $DB | grep attn4 | head -2 | cut -c1-160 >> $O/pmc_summary.txt 2>&1
done
rm -rf $O/pmc_*/
cat $O/pmc_summary.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "prescaled or attn4" > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout 600 python bench.py --steps 3 --warmup 1 > $O/bench.log 2>&1; grep -o '"ms_per_step": [0-9.]*\|"value": [0-9.]*\|"achieved": [0-9.]*\|"step_mfma_frac": [0-9.]*\|"ms_per_launch": [0-9.]*\|"traffic": [a-z0-9.]*' $O/bench.log | head -8
