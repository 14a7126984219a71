#!/bin/bash
# round 3, call m: conv4 in the product path: VAE tests (product + variants), probe, bench
O=gpurun_out/r03m
mkdir -p $O
timeout 1500 python -m pytest tests/test_vae_gpu.py -q --durations=5 > $O/vae_tests.log 2>&1; tail -8 $O/vae_tests.log
SCAIL_ABLATIONS=1 timeout 900 python -m pytest tests/test_vae_gpu.py -q -k "causal_conv3d or resample" > $O/vae_tests_variants.log 2>&1; tail -3 $O/vae_tests_variants.log
SCAIL_ABLATIONS=1 timeout 600 python tools/conv4_probe.py --variants "" > $O/conv4_probe.log 2>&1; grep -v '"check"' $O/conv4_probe.log | cut -c1-200
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.log 2>&1; tail -c 1500 $O/bench.log
