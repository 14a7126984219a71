set -x
O=gpurun_out/r02q
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_shape_probe.hip -o /tmp/mfma_shape_probe 2>/dev/null && timeout 120 /tmp/mfma_shape_probe > $O/mfma_shape.log 2>&1; cat $O/mfma_shape.log
SCAIL_ABLATIONS=1 timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_vae_gpu.py -m gpu -x -q > $O/gpu_tests_variants.log 2>&1; tail -4 $O/gpu_tests_variants.log
