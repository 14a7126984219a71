set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02a; mkdir -p $O
SCAIL_ABLATIONS=1 timeout 900 python tools/attn4_tune.py --skip-check --variants ",m16c2la2" --ablations "m16_abl_fma,m16_abl_fma_c3,m16_abl_fma_sm44,m16_abl_fma_max,abl_fma" > $O/attn_m16b.log 2>&1; grep -c '"ok": true' $O/attn_m16b.log; grep '"ok": false' $O/attn_m16b.log | head -5; grep kernel $O/attn_m16b.log | cut -c1-200
