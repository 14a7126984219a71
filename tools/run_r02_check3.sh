set -x
O=gpurun_out/r02s
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_dit_gpu.py -m gpu -x -q -k "sequence_parallel or sp" > $O/sp_tests.log 2>&1; tail -4 $O/sp_tests.log
timeout 600 python tools/sp_rank_compute.py 1 2 4 8 > $O/sp_rank_compute.log 2>&1; tail -5 $O/sp_rank_compute.log
