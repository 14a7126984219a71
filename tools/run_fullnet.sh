set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02net; mkdir -p $O
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q -s -k six_layer > $O/six_layer.log 2>&1; grep -E "layer|velocity|passed|failed|Error" $O/six_layer.log | head -14
