set -x
O=gpurun_out/r03e
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
SCAIL_ABLATIONS=1 timeout 900 python -m pytest tests -m "gpu and variant" -q > $O/gpu_tests_variants.log 2>&1; tail -3 $O/gpu_tests_variants.log
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -q -k "gemm or block or c_step or six_layer" > $O/gemm_tests.log 2>&1; tail -3 $O/gemm_tests.log
timeout 600 python tools/gemm_probe.py > $O/gemm_probe.log 2>&1; cat $O/gemm_probe.log | cut -c1-300
SCAIL_ABLATIONS=1 timeout 600 python tools/attn_sp_shape_probe.py 8 4 1 > $O/attn_sp_shape.log 2>&1; cat $O/attn_sp_shape.log
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_sp8 -o sp8 -- python tools/sp_rank_compute.py 8 > $O/sp8_prof.log 2>&1
DB=$(find $O/prof_sp8 -name "*.db" | head -1); python tools/rocpd_summary.py $DB > $O/sp8_kernel_stats.md 2>&1; head -24 $O/sp8_kernel_stats.md | cut -c1-150
rm -rf $O/prof_sp8
