set -x
O=gpurun_out/r03j
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --durations=5 > $O/gpu_tests.log 2>&1; tail -10 $O/gpu_tests.log
SCAIL_ABLATIONS=1 timeout 900 python -m pytest tests -m "gpu and variant" -q > $O/gpu_tests_variants.log 2>&1; tail -3 $O/gpu_tests_variants.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python tools/gemm_probe.py > $O/gemm_probe.log 2>&1; cat $O/gemm_probe.log | cut -c1-300
rm -f $O/pmc_summary.txt
for C in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/pmc_$N -o pmc -- python tools/attn_pmc_probe.py prescaled 3 > $O/pmc_$N.log 2>&1
  DB=$(find $O/pmc_$N -name "*.db" | head -1); python tools/rocpd_counters.py $DB attn4 >> $O/pmc_summary.txt 2>&1
  python tools/rocpd_summary.py $DB | grep attn4 | head -2 | cut -c1-160 >> $O/pmc_summary.txt 2>&1
  rm -rf $O/pmc_$N
done
for C in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/pmcg_$N -o pmc -- python tools/gemm_layer_pmc_probe.py 2 > $O/pmcg_$N.log 2>&1
  DB=$(find $O/pmcg_$N -name "*.db" | head -1); python tools/rocpd_counters.py $DB gemm4 >> $O/pmc_summary.txt 2>&1
  python tools/rocpd_summary.py $DB | grep gemm4 | head -5 | cut -c1-160 >> $O/pmc_summary.txt 2>&1
  rm -rf $O/pmcg_$N
done
for CC in 96 192; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/pmcc_${CC}_$C -o pmc -- python tools/conv_pmc_probe.py $CC 2 > $O/pmcc_${CC}_$C.log 2>&1
    DB=$(find $O/pmcc_${CC}_$C -name "*.db" | head -1); python tools/rocpd_counters.py $DB conv_halo | awk -v C=$CC '{print "conv C=" C, $(NF-4), $(NF-2)}' >> $O/pmc_summary.txt 2>&1
    rm -rf $O/pmcc_${CC}_$C
  done
done
grep "FETCH\|WRITE" $O/pmc_summary.txt
python tools/update_traffic.py $O/pmc_summary.txt "round 3 final tree (profiles/r03_pmc_attn4_gemm4_raw.txt, tools/run_r03_j.sh)" > $O/traffic_update.log 2>&1; cp profiles/traffic.json $O/traffic.json; tail -3 $O/traffic_update.log
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_prof.log 2>&1
DB=$(find $O/prof_bench -name "*.db" | head -1); python tools/rocpd_summary.py $DB > $O/bench_kernel_stats.md 2>&1; head -12 $O/bench_kernel_stats.md | cut -c1-150
rm -rf $O/prof_bench
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench.log 2>&1; tail -c 6000 $O/bench.log | cut -c1-3500
timeout 600 python bench.py --latent-hw 60 104 --steps 2 --warmup 1 --no-vae --no-cpu-baseline > $O/bench_480x832.log 2>&1; grep -o '"ms_per_step": [0-9.]*\|"achieved": [0-9.]*' $O/bench_480x832.log
