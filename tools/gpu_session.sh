#!/bin/bash
# One parameterised GPU session (replaces the per-call tools/run_r0*_*.sh one-shots of rounds 1-3; those exact command lines are in the
# git history next to the profiles/ logs they produced).
#   usage (on the GPU box, through gpurun):  bash tools/gpu_session.sh <tag> <leg> [<leg> ...]
#   output: gpurun_out/<tag>/...   legs run in the order given; every leg is bounded by its own `timeout`.
# legs:
#   tests            pytest -m gpu (product build)                      tests-x        same, stop at the first failure
#   variants         pytest -m "gpu and variant" on the measurement build (SCAIL_ABLATIONS=1)
#   smoke            __graft_entry__.smoke()
#   bench            python bench.py (driver style, default K / W)      bench-quick    --steps 3 --warmup 1 --no-cpu-baseline
#   bench-prof       rocprofv3 --kernel-trace --stats of bench-quick -> bench_kernel_stats.md
#   bench-2char      config 5 extension line                            bench-480      480x832x81f line
#   vae-prof         kernel stats of the VAE leg (tools/vae_leg_probe.py)
#   pmc-attn | pmc-gemm | pmc-conv    PMC passes (FETCH_SIZE, WRITE_SIZE, busy counters; each its own rocprofv3 run) -> pmc_summary.txt
#   pmc-s2           FETCH_SIZE / WRITE_SIZE of Resample's stride-2 convolutions, gather kernel vs conv_s2_kernel -> pmc_s2_summary.txt
#   traffic          tools/update_traffic.py from pmc_summary.txt (run after the pmc legs)
#   sp               tools/sp_rank_compute.py 1 2 4 8                   sp8-prof | sp4-prof   kernel stats of one rank's launches of an 8- / 4-rank step
#   py:<script> [..] python tools/<script> (arguments up to the next leg name are NOT supported: wrap them in quotes: "py:gemm_probe.py 4")
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p "$O"
BUSY="GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
LDS="SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS"
pmc_pass() {   # pmc_pass <name> <kernel filter> <counters> -- <command...>
  local name=$1 filt=$2 ctrs=$3; shift 4
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs -d "$O/pmc_$name" -o pmc -- "$@" > "$O/pmc_$name.log" 2>&1
  local db; db=$(find "$O/pmc_$name" -name "*.db" | head -1)
  python tools/rocpd_counters.py "$db" "$filt" >> "$O/pmc_summary.txt" 2>&1
  python tools/rocpd_summary.py "$db" | grep "$filt" | head -5 | cut -c1-160 >> "$O/pmc_summary.txt" 2>&1
  rm -rf "$O/pmc_$name"
}
kstats() {     # kstats <name> <command...>: kernel trace -> <name>_kernel_stats.md
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats -d "$O/prof_$name" -o "$name" -- "$@" > "$O/${name}_prof.log" 2>&1
  local db; db=$(find "$O/prof_$name" -name "*.db" | head -1)
  python tools/rocpd_summary.py "$db" > "$O/${name}_kernel_stats.md" 2>&1; head -16 "$O/${name}_kernel_stats.md" | cut -c1-150
  rm -rf "$O/prof_$name"
}
for LEG in "$@"; do
  echo "==== leg: $LEG"
  case "$LEG" in
    tests)       timeout 1800 python -m pytest tests -m gpu -q --durations=8 > "$O/gpu_tests.log" 2>&1; tail -14 "$O/gpu_tests.log" ;;
    tests-x)     timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 > "$O/gpu_tests.log" 2>&1; tail -30 "$O/gpu_tests.log" ;;
    variants)    SCAIL_ABLATIONS=1 timeout 900 python -m pytest tests -m "gpu and variant" -q > "$O/gpu_tests_variants.log" 2>&1; tail -3 "$O/gpu_tests_variants.log" ;;
    smoke)       timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$O/smoke.log" 2>&1; tail -2 "$O/smoke.log" ;;
    bench)       timeout 1200 python bench.py > "$O/bench.log" 2>&1; tail -c 7000 "$O/bench.log" | cut -c1-5000 ;;
    bench-quick) timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$O/bench_quick.log" 2>&1; tail -c 5000 "$O/bench_quick.log" | cut -c1-4000 ;;
    bench-prof)  kstats bench python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vae --no-extra-legs --no-pmc ;;     # (the legs after the timed region launch the same kernels at other shapes)
    bench-2char) timeout 600 python bench.py --config 14b-2char --steps 2 --warmup 1 --no-vae --no-cpu-baseline > "$O/bench_2char.log" 2>&1; grep -o '"ms_per_step": [0-9.]*\|"achieved": [0-9.]*' "$O/bench_2char.log" ;;
    bench-480)   timeout 600 python bench.py --latent-hw 60 104 --steps 2 --warmup 1 --no-vae --no-cpu-baseline > "$O/bench_480x832.log" 2>&1; grep -o '"ms_per_step": [0-9.]*\|"achieved": [0-9.]*' "$O/bench_480x832.log" ;;
    vae-prof)    timeout 900 rocprofv3 --kernel-trace --stats -d "$O/prof_vae" -o vae -- python tools/vae_leg_probe.py > "$O/vae_prof.log" 2>&1
                 DB=$(find "$O/prof_vae" -name "*.db" | head -1); python tools/rocpd_summary.py "$DB" --by-grid > "$O/vae_kernel_stats.md" 2>&1; head -24 "$O/vae_kernel_stats.md" | cut -c1-150; rm -rf "$O/prof_vae" ;;
    pmc-attn)    for C in FETCH_SIZE WRITE_SIZE "$BUSY"; do pmc_pass "attn_$(echo $C | cut -d' ' -f1)" attn4 "$C" -- python tools/attn_pmc_probe.py prescaled 3; done; grep -h "FETCH\|WRITE" "$O/pmc_summary.txt" | cut -c1-200 ;;
    pmc-gemm)    for C in FETCH_SIZE WRITE_SIZE "$BUSY" "$LDS"; do pmc_pass "gemm_$(echo $C | cut -d' ' -f1)" gemm4 "$C" -- python tools/gemm_layer_pmc_probe.py 2; done ;;
    pmc-conv)    for CC in 96 192 384; do for C in FETCH_SIZE WRITE_SIZE; do
                   timeout 300 rocprofv3 --kernel-trace --pmc $C -d "$O/pmcc_${CC}_$C" -o pmc -- python tools/conv_pmc_probe.py $CC 2 > "$O/pmcc_${CC}_$C.log" 2>&1
                   DB=$(find "$O/pmcc_${CC}_$C" -name "*.db" | head -1); python tools/rocpd_counters.py "$DB" conv4 | awk -v C=$CC '{print "conv4 C=" C, $(NF-4), $(NF-2)}' >> "$O/pmc_summary.txt" 2>&1
                   rm -rf "$O/pmcc_${CC}_$C"; done; done
                 for C in "$BUSY" "$LDS"; do pmc_pass "conv_$(echo $C | cut -d' ' -f1)" conv4 "$C" -- python tools/conv_pmc_probe.py 96 2; done ;;
    pmc-s2)      for M in 0 1; do for C in FETCH_SIZE WRITE_SIZE; do      # per-grid means: the two shapes are different grids of the same kernel
                   timeout 300 rocprofv3 --kernel-trace --pmc $C -d "$O/pmcs2_${M}_$C" -o pmc -- python tools/conv_s2_pmc_probe.py $M > "$O/pmcs2_${M}_$C.log" 2>&1
                   DB=$(find "$O/pmcs2_${M}_$C" -name "*.db" | head -1); echo "conv_s2 option $M, $C (KiB per launch; x 2 for fetches: the guide's gfx950 correction)" >> "$O/pmc_s2_summary.txt"
                   python tools/rocpd_counters.py "$DB" conv_ --by-grid | cut -c1-170 >> "$O/pmc_s2_summary.txt" 2>&1; rm -rf "$O/pmcs2_${M}_$C"; done; done; cat "$O/pmc_s2_summary.txt" ;;
    traffic)     python tools/update_traffic.py "$O/pmc_summary.txt" "$TAG (tools/gpu_session.sh)" > "$O/traffic_update.log" 2>&1; cp profiles/traffic.json "$O/traffic.json"; tail -30 "$O/traffic_update.log" ;;
    sp8-prof)    kstats sp8 python tools/sp_rank_compute.py 8 ;;
    sp4-prof)    kstats sp4 python tools/sp_rank_compute.py 4 ;;
    sp)          timeout 900 python tools/sp_rank_compute.py 1 2 4 8 > "$O/sp_rank_compute.log" 2>&1; cut -c1-220 "$O/sp_rank_compute.log" ;;
    py:*)        S=${LEG#py:}; N=$(echo "$S" | cut -d' ' -f1 | sed 's/\.py$//'); timeout 900 python tools/$S > "$O/$N.log" 2>&1; tail -40 "$O/$N.log" | cut -c1-300 ;;
    abl:*)       S=${LEG#abl:}; N=$(echo "$S" | cut -d' ' -f1 | sed 's/\.py$//'); SCAIL_ABLATIONS=1 timeout 900 python tools/$S > "$O/${N}_abl.log" 2>&1; tail -40 "$O/${N}_abl.log" | cut -c1-300 ;;
    *)           echo "unknown leg $LEG" ;;
  esac
done
