set -x
O=gpurun_out/r02p
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "cross_attn2 or attn4" > $O/cross_tests.log 2>&1; tail -3 $O/cross_tests.log
timeout 200 python tools/cross_attn_probe.py > $O/cross_probe.log 2>&1; tail -2 $O/cross_probe.log
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
SCAIL_ABLATIONS=1 timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_vae_gpu.py -m gpu -x -q > $O/gpu_tests_variants.log 2>&1; tail -4 $O/gpu_tests_variants.log
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_vendor -o vendor -- python tools/vendor_gemm_probe.py > $O/vendor.log 2>&1; tail -5 $O/vendor.log
DB=$(find $O/prof_vendor -name "*.db" | head -1); python tools/rocpd_summary.py $DB > $O/vendor_kernel_stats.md 2>&1; head -12 $O/vendor_kernel_stats.md | cut -c1-400
rm -rf $O/prof_vendor
timeout 600 python bench.py --steps 3 --warmup 1 > $O/bench.log 2>&1; tail -c 2500 $O/bench.log
