set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02sp; mkdir -p $O
timeout 600 python tools/e2e_sp_emulated.py 8 ulysses 2 > $O/sp8.log 2>&1; tail -1 $O/sp8.log | cut -c1-400
timeout 600 python tools/e2e_sp_emulated.py 4 ulysses 2 > $O/sp4.log 2>&1; tail -1 $O/sp4.log | cut -c1-400
timeout 600 python tools/e2e_sp_emulated.py 2 allgather 2 > $O/sp2.log 2>&1; tail -1 $O/sp2.log | cut -c1-400
