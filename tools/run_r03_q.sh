#!/bin/bash
# round 3, call q: whole-line (LDS-staged) GEMM epilogue vs the shipped one, same process
O=gpurun_out/r03q
mkdir -p $O
export SCAIL_ABLATIONS=1
timeout 900 python tools/gemm_pst_probe.py stg > $O/gemm_stg.log 2>&1; cut -c1-330 $O/gemm_stg.log
timeout 900 python tools/gemm_pst_probe.py stgnt > $O/gemm_stgnt.log 2>&1; grep 97664 $O/gemm_stgnt.log | cut -c1-330
