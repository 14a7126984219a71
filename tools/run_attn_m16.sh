set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02a; mkdir -p $O
SCAIL_ABLATIONS=1 timeout 900 python tools/attn4_tune.py --skip-check --prescaled --variants "m16g_c1,m16g_c1sm44,m16g_c1sm60,m16g_c1la1,m16g_c1la4,m16g_c1dmamid,m16g_c1dmaspread,m16f_c1,m16g_c1" --ablations "" > $O/attn_m16h.log 2>&1; grep kernel $O/attn_m16h.log | cut -c1-170
