#!/bin/bash
# round 3, call l: PMC counters of the generated conv4 kernel (C = 192 and 96 shapes)
O=gpurun_out/r03l
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export SCAIL_ABLATIONS=1
rm -f $O/pmc_summary.txt
for SH in 192 96; do
for C in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM"; do
  N=$(echo $C | cut -d' ' -f1)
  echo "== C=$SH counters: $C" >> $O/pmc_summary.txt
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/pmc_$N -o pmc -- python tools/conv_pmc_probe.py $SH 2 conv4 > $O/pmc_$N.log 2>&1
  DB=$(find $O/pmc_$N -name "*.db" | head -1); python tools/rocpd_counters.py $DB conv4 >> $O/pmc_summary.txt 2>&1
  python tools/rocpd_summary.py $DB | grep conv4 | head -2 | cut -c1-160 >> $O/pmc_summary.txt 2>&1
  rm -rf $O/pmc_$N
done
done
cat $O/pmc_summary.txt | cut -c1-200
