set -x
O=gpurun_out/r02rg
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "prescaled or attn4" > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 600 python bench.py --steps 3 --warmup 1 > $O/bench.log 2>&1; grep -o '"ms_per_step": [0-9.]*\|"achieved": [0-9.]*\|"ms_per_launch": [0-9.]*\|"traffic": [a-z0-9.]*' $O/bench.log | head -5
