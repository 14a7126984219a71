set -x
O=gpurun_out/r02fin
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python bench.py --steps 3 --warmup 1 > $O/bench_prof.log 2>&1
DB=$(find $O/prof_bench -name "*.db" | head -1); python tools/rocpd_summary.py $DB --split flash_attn:20000 > $O/bench_kernel_stats.md 2>&1; head -12 $O/bench_kernel_stats.md | cut -c1-150
rm -rf $O/prof_bench
timeout 600 python bench.py > $O/bench.log 2>&1; grep -o '"ms_per_step": [0-9.]*\|"value": [0-9.]*\|"achieved": [0-9.]*\|"steps": [0-9]*\|"ms_per_launch": [0-9.]*\|"traffic": [a-z0-9.]*' $O/bench.log | head -8
