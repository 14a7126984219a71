#!/bin/bash
# round 3, call p: conv4 tile walk (runs for one n tile, strided for several): timing + fabric traffic of the three shapes + VAE leg
O=gpurun_out/r03p
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
SCAIL_ABLATIONS=1 timeout 600 python tools/conv4_probe.py --variants "" > $O/conv4_probe.log 2>&1; grep -v '"check"' $O/conv4_probe.log | cut -c1-200
rm -f $O/pmc_summary.txt
for CC in 96 192 384; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/pmcc_${CC}_$C -o pmc -- python tools/conv_pmc_probe.py $CC 2 > $O/pmcc_${CC}_$C.log 2>&1
    DB=$(find $O/pmcc_${CC}_$C -name "*.db" | head -1); python tools/rocpd_counters.py $DB conv4 | awk -v C=$CC '{print "conv4 C=" C, $(NF-4), $(NF-2)}' >> $O/pmc_summary.txt 2>&1
    rm -rf $O/pmcc_${CC}_$C
  done
done
cat $O/pmc_summary.txt
python tools/update_traffic.py $O/pmc_summary.txt "round 3, generated convolution kernels (profiles/r03_pmc_conv4_raw.txt, tools/run_r03_p.sh)" > $O/traffic_update.log 2>&1; cp profiles/traffic.json $O/traffic.json
timeout 900 python -m pytest tests/test_vae_gpu.py -q 2>&1 | tail -2
timeout 600 python tools/vae_leg_probe.py 2>/dev/null | grep -o '"encode_ms": [0-9.]*\|"decode_ms": [0-9.]*\|"conv_traffic": .*'
