set -x
O=gpurun_out/r02end
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --steps 10 --warmup 2 > $O/bench10.log 2>&1; grep -o '"ms_per_step": [0-9.]*\|"value": [0-9.]*\|"achieved": [0-9.]*\|"ms_per_launch": [0-9.]*' $O/bench10.log | head -4
timeout 600 python tools/sp_rank_compute.py 1 2 4 8 > $O/sp_rank_compute.log 2>&1; tail -4 $O/sp_rank_compute.log
timeout 900 python tools/e2e_fullsize.py > $O/e2e_fullsize.log 2>&1; tail -3 $O/e2e_fullsize.log | cut -c1-300
