#!/bin/bash
# round 3, call s: the bench legs on the final tree (driver style, config 5, 480x832) + SP compute-side efficiency
O=gpurun_out/r03s
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python bench.py --steps 10 --warmup 3 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-1200
timeout 600 python bench.py --config 14b-2char --steps 2 --warmup 1 --no-vae --no-cpu-baseline > $O/bench_2char.log 2>&1; grep -o '"ms_per_step": [0-9.]*\|"achieved": [0-9.]*' $O/bench_2char.log
timeout 600 python bench.py --latent-hw 60 104 --steps 2 --warmup 1 --no-vae --no-cpu-baseline > $O/bench_480x832.log 2>&1; grep -o '"ms_per_step": [0-9.]*\|"achieved": [0-9.]*' $O/bench_480x832.log
timeout 900 python tools/sp_rank_compute.py 1 2 4 8 > $O/sp_rank_compute.log 2>&1; cat $O/sp_rank_compute.log | cut -c1-200
