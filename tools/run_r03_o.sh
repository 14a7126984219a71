#!/bin/bash
# round 3, call o: validation of the tree with the generated convolution kernels and the staged GEMM epilogue: GPU suites (product + measurement build), smoke, conv4 PMC
# traffic (-> profiles/traffic.json, stamped with the blob of csrc/conv4.s), VAE leg kernel stats, bench (driver style) + its kernel stats
set -x
O=gpurun_out/r03o
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --durations=5 > $O/gpu_tests.log 2>&1; tail -10 $O/gpu_tests.log
SCAIL_ABLATIONS=1 timeout 900 python -m pytest tests -m "gpu and variant" -q > $O/gpu_tests_variants.log 2>&1; tail -3 $O/gpu_tests_variants.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
rm -f $O/pmc_summary.txt
for CC in 96 192 384; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/pmcc_${CC}_$C -o pmc -- python tools/conv_pmc_probe.py $CC 2 > $O/pmcc_${CC}_$C.log 2>&1
    DB=$(find $O/pmcc_${CC}_$C -name "*.db" | head -1); python tools/rocpd_counters.py $DB conv4 | awk -v C=$CC '{print "conv4 C=" C, $(NF-4), $(NF-2)}' >> $O/pmc_summary.txt 2>&1
    rm -rf $O/pmcc_${CC}_$C
  done
done
for C in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/pmcg_$N -o pmc -- python tools/gemm_layer_pmc_probe.py 2 > $O/pmcg_$N.log 2>&1
  DB=$(find $O/pmcg_$N -name "*.db" | head -1); python tools/rocpd_counters.py $DB gemm4 >> $O/pmc_summary.txt 2>&1
  python tools/rocpd_summary.py $DB | grep gemm4 | head -5 | cut -c1-160 >> $O/pmc_summary.txt 2>&1
  rm -rf $O/pmcg_$N
done
for C in "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/pmcs_$N -o pmc -- python tools/conv_pmc_probe.py 96 2 > $O/pmcs_$N.log 2>&1
  DB=$(find $O/pmcs_$N -name "*.db" | head -1); python tools/rocpd_counters.py $DB conv4 >> $O/pmc_summary.txt 2>&1
  python tools/rocpd_summary.py $DB | grep conv4 | head -1 | cut -c1-160 >> $O/pmc_summary.txt 2>&1
  rm -rf $O/pmcs_$N
done
cat $O/pmc_summary.txt | cut -c1-160
python tools/update_traffic.py $O/pmc_summary.txt "round 3 final tree: generated convolution kernels, staged GEMM epilogue (profiles/r03_pmc_final_raw.txt, tools/run_r03_o.sh)" > $O/traffic_update.log 2>&1; cp profiles/traffic.json $O/traffic.json; tail -30 $O/traffic_update.log
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_vae -o vae -- python tools/vae_leg_probe.py > $O/vae_prof.log 2>&1
DB=$(find $O/prof_vae -name "*.db" | head -1); python tools/rocpd_summary.py $DB --by-grid > $O/vae_kernel_stats.md 2>&1; head -14 $O/vae_kernel_stats.md | cut -c1-150
rm -rf $O/prof_vae
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_prof.log 2>&1
DB=$(find $O/prof_bench -name "*.db" | head -1); python tools/rocpd_summary.py $DB > $O/bench_kernel_stats.md 2>&1; head -12 $O/bench_kernel_stats.md | cut -c1-150
rm -rf $O/prof_bench
SCAIL_ABLATIONS=1 timeout 600 python tools/gemm_pst_probe.py part > $O/gemm_part_ab.log 2>&1; grep 97664 $O/gemm_part_ab.log | cut -c1-300
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.log 2>&1; tail -c 6000 $O/bench.log | cut -c1-4000
