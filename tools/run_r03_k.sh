#!/bin/bash
# round 3, call k: the generated conv4 kernel (check + A/B vs the hipcc halo kernel + phase timers) + VAE tests + VAE leg
mkdir -p gpurun_out
export SCAIL_ABLATIONS=1
timeout 600 python tools/conv4_probe.py --variants "" > gpurun_out/r03_conv4_probe.log 2>&1
echo "exit $?" >> gpurun_out/r03_conv4_probe.log
timeout 300 python tools/conv4_probe.py --prof --skip-check >> gpurun_out/r03_conv4_probe.log 2>&1
echo "exit $?" >> gpurun_out/r03_conv4_probe.log
grep -v '"check"' gpurun_out/r03_conv4_probe.log | cut -c1-600
grep -c '"ok": true' gpurun_out/r03_conv4_probe.log; grep -c '"ok": false' gpurun_out/r03_conv4_probe.log
unset SCAIL_ABLATIONS
timeout 900 python -m pytest tests/test_vae_gpu.py -q 2>&1 | tail -2
timeout 600 python tools/vae_leg_probe.py 2>/dev/null | grep -o '"encode_ms": [0-9.]*\|"decode_ms": [0-9.]*'
true
