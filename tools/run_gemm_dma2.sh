set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02g; mkdir -p $O
SCAIL_ABLATIONS=1 timeout 600 python tools/gemm4_tune.py --skip-check --vendor --variants "," --ablations "abl_wpack,abl_xpack,abl_wpack_xpack" > $O/gemm_pack.log 2>&1; grep qkv $O/gemm_pack.log | cut -c1-900
