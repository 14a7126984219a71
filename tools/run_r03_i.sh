set -x
O=gpurun_out/r03i
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
SCAIL_ABLATIONS=1 timeout 600 python tools/conv_probe.py --knobs 4,5 > $O/conv_pf.log 2>&1; cat $O/conv_pf.log | cut -c1-200
