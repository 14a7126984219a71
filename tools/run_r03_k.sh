#!/bin/bash
# round 3, call k: first GPU run of the generated conv4 kernel (check + A/B vs the hipcc halo kernel)
mkdir -p gpurun_out
export SCAIL_ABLATIONS=1
timeout 600 python tools/conv4_probe.py --variants ",abl_dma,abl_lds,abl_bar,abl_dma_lds,abl_patch,c2,rd2,rd6,p30" > gpurun_out/r03_conv4_probe.log 2>&1
echo "exit $?" >> gpurun_out/r03_conv4_probe.log
tail -50 gpurun_out/r03_conv4_probe.log
