set -x
O=gpurun_out/r03h
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
SCAIL_ABLATIONS=1 timeout 900 python tools/gemm_pst_probe.py > $O/gemm_pst.log 2>&1; cat $O/gemm_pst.log | cut -c1-330
