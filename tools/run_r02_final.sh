set -x
O=gpurun_out/r02z
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python bench.py --steps 3 --warmup 1 > $O/bench_prof.log 2>&1; tail -c 1800 $O/bench_prof.log
DB=$(find $O/prof_bench -name "*.db" | head -1); python tools/rocpd_summary.py $DB --split flash_attn:20000 > $O/bench_kernel_stats.md 2>&1; head -24 $O/bench_kernel_stats.md | cut -c1-180
rm -rf $O/prof_bench
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
