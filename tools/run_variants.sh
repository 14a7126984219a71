set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02var; mkdir -p $O
SCAIL_ABLATIONS=1 timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_vae_gpu.py -m gpu -q > $O/gpu_tests_variants.log 2>&1; tail -3 $O/gpu_tests_variants.log
