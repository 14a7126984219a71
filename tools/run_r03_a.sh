set -x
O=gpurun_out/r03a
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -x --durations=15 > $O/gpu_tests.log 2>&1; tail -25 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench.log 2>&1; tail -c 6000 $O/bench.log
V=",m16f_noopt,m16f_opt_a6,m16f_opt_a8,m16f_opt_sm36,m16f_opt_sm48,m16f_opt_sm50,m16f_opt_qb_sm44,m16f_opt_qb_sm52,m16f_opt_qb_sm60,m16f_opt_c2,m16f_opt_la1,m16f_opt_la4,m16f_opt_dmamid,m16f_opt_dmaspread,m16f_u_c1,m16f_u_c1le2,m16f_u_c1le4,m16f_u_c1le2sm50,m16f_u_c2,m16f_u_c2la2,m16f_le2,m16f_qb_le2,,m16f_noopt"
A="m16f_opt_abl_dma,m16f_opt_abl_lds,m16f_opt_abl_valu,m16f_opt_abl_bar,m16f_opt_abl_dma_lds_valu"
SCAIL_ABLATIONS=1 timeout 900 python tools/attn4_tune.py --prescaled --skip-check --heads 8 --iters 5 --variants "$V" --ablations "$A" --full --full-also "default,m16f_noopt" > $O/attn_variants.log 2>&1; cat $O/attn_variants.log | cut -c1-200
SCAIL_ABLATIONS=1 timeout 600 python tools/gemm_table_probe.py > $O/gemm_table.log 2>&1; cat $O/gemm_table.log | cut -c1-200
