set -x
O=gpurun_out/r02h
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm" > $O/gemm_tests.log 2>&1; tail -3 $O/gemm_tests.log
timeout 300 python tools/gemm_probe.py > $O/gemm_probe.log 2>&1; tail -4 $O/gemm_probe.log | cut -c1-400
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 600 python bench.py --steps 3 --warmup 1 > $O/bench.log 2>&1; grep -o '"ms_per_step": [0-9.]*\|"value": [0-9.]*\|"achieved": [0-9.]*\|"step_mfma_frac": [0-9.]*' $O/bench.log | head -5
SCAIL_ABLATIONS=1 timeout 600 python tools/gemm4_tune.py --skip-check --variants ",mi16_early175,mi16_early19,mi16_early185,mi16_early175c2,mi16_d15" --ablations "" > $O/gemm_mi16d.log 2>&1; tail -4 $O/gemm_mi16d.log | cut -c1-900
