set -x
O=gpurun_out/r03c
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -q -k "gemm or block or c_step or six_layer" > $O/gemm_tests.log 2>&1; tail -4 $O/gemm_tests.log
timeout 600 python tools/gemm_probe.py > $O/gemm_probe.log 2>&1; cat $O/gemm_probe.log | cut -c1-300
V=",m16f_q_sm40,m16f_q_sm48,m16f_q_le4,m16f_q_le8,m16f_q_v20,m16f_q_v30,m16f_q_k16,m16f_q_k24s25,m16f_q_k20s30,m16f_q_dmamid,m16f_q_dmaspread,m16f_q_dmalate,m16f_q_la1,m16f_q_la4,m16f_q_u_le4,m16f_opt_db,,m16f_noopt"
SCAIL_ABLATIONS=1 timeout 900 python tools/attn4_tune.py --prescaled --skip-check --heads 8 --iters 7 --variants "$V" --ablations "" > $O/attn_variants2.log 2>&1; cat $O/attn_variants2.log | cut -c1-160
