set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02g; mkdir -p $O
SCAIL_ABLATIONS=1 timeout 600 python tools/gemm4_tune.py --skip-check --vendor > $O/gemm.log 2>&1; tail -5 $O/gemm.log | cut -c1-1500
