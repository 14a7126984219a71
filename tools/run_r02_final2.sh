set -x
O=gpurun_out/r02z
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
timeout 600 python bench.py --steps 3 --warmup 1 > $O/bench.log 2>&1; tail -c 600 $O/bench.log
