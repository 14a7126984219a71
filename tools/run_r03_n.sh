#!/bin/bash
# round 3, call n: kernel stats of the VAE leg with the generated convolution kernels + the VAE tests
O=gpurun_out/r03n
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_vae_gpu.py -q > $O/vae_tests.log 2>&1; tail -3 $O/vae_tests.log
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_vae -o vae -- python tools/vae_leg_probe.py > $O/vae_prof.log 2>&1
DB=$(find $O/prof_vae -name "*.db" | head -1); python tools/rocpd_summary.py $DB --by-grid > $O/vae_kernel_stats.md 2>&1; head -16 $O/vae_kernel_stats.md | cut -c1-170
rm -rf $O/prof_vae
grep -o '"encode_ms": [0-9.]*\|"decode_ms": [0-9.]*' $O/vae_prof.log
timeout 600 python tools/vae_leg_probe.py 2>/dev/null | grep -o '"encode_ms": [0-9.]*\|"decode_ms": [0-9.]*'
