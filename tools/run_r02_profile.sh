set -x
mkdir -p gpurun_out/r02f
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02f/gpu_tests.log 2>&1; tail -4 gpurun_out/r02f/gpu_tests.log
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/r02f/prof_bench -o bench -- python bench.py --steps 3 --warmup 1 > gpurun_out/r02f/bench_prof.log 2>&1; tail -c 1500 gpurun_out/r02f/bench_prof.log
DB=$(find gpurun_out/r02f/prof_bench -name "*.db" | head -1); python tools/rocpd_summary.py $DB --split flash_attn:20000 > gpurun_out/r02f/bench_kernel_stats.md 2>&1; head -25 gpurun_out/r02f/bench_kernel_stats.md
for C in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d gpurun_out/r02f/pmc_$N -o pmc -- python tools/microbench.py attn --heads 40 --iters 2 > gpurun_out/r02f/pmc_$N.log 2>&1
  DB=$(find gpurun_out/r02f/pmc_$N -name "*.db" | head -1); python tools/rocpd_counters.py $DB attn4 >> gpurun_out/r02f/pmc_summary.txt 2>&1
done
cat gpurun_out/r02f/pmc_summary.txt
rm -rf gpurun_out/r02f/prof_bench gpurun_out/r02f/pmc_*/
