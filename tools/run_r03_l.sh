#!/bin/bash
# round 3, call l: fabric traffic of the generated conv4 kernel, default vs non-temporal stores (C = 96 shape) + timing
O=gpurun_out/r03l
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export SCAIL_ABLATIONS=1
rm -f $O/pmc_summary.txt
for K in conv4 conv4:nt; do
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== $K C=96 counter $C" >> $O/pmc_summary.txt
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/pmc_$C -o pmc -- python tools/conv_pmc_probe.py 96 2 $K > $O/pmc_$C.log 2>&1
  DB=$(find $O/pmc_$C -name "*.db" | head -1); python tools/rocpd_counters.py $DB conv4 >> $O/pmc_summary.txt 2>&1
  rm -rf $O/pmc_$C
done
done
cat $O/pmc_summary.txt | cut -c1-200
timeout 600 python tools/conv4_probe.py --skip-check --variants ",nt" 2>&1 | grep -v resid | cut -c1-200
