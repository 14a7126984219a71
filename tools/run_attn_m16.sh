set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02a; mkdir -p $O
SCAIL_ABLATIONS=1 timeout 900 python tools/attn4_tune.py --skip-check --prescaled --iters 5 --variants ",m16g_c1sm44,,m16g_c1sm44" --ablations "" > $O/attn_ragged_ab.log 2>&1; grep kernel $O/attn_ragged_ab.log | cut -c1-170
