set -x
O=gpurun_out/r03g
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python tools/gemm_epilogue_probe.py > $O/gemm_epilogue.log 2>&1; cat $O/gemm_epilogue.log | cut -c1-250
timeout 300 python tools/gemm_probe.py > $O/gemm_probe.log 2>&1; cat $O/gemm_probe.log | cut -c1-200
