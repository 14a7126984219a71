set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02g; mkdir -p $O
SCAIL_ABLATIONS=1 timeout 600 python tools/gemm4_tune.py --skip-check --vendor --variants ",mi16_d125,mi16_d15,mi16_early175,mi16_d125nt,mi16_d125la2,mi16_early" --ablations "" > $O/gemm_mi16c.log 2>&1; tail -4 $O/gemm_mi16c.log | cut -c1-1600
