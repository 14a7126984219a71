set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r02v; mkdir -p $O
for C in "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d $O/p_$N -o pmc -- python tools/vendor_one_gemm.py 3 > $O/log_$N.txt 2>&1
  DB=$(find $O/p_$N -name "*.db" | head -1)
  python tools/rocpd_counters.py $DB Cijk >> $O/summary.txt 2>&1
  python tools/rocpd_summary.py $DB | grep -i Cijk | head -2 | cut -c1-200 >> $O/summary.txt 2>&1
done
python - <<'PY' >> $O/summary.txt 2>&1
import sqlite3, glob
for db in glob.glob("gpurun_out/r02v/p_FETCH_SIZE/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    for t in tabs:
        if "kernel" in t.lower() and "symbol" in t.lower():
            cols = [r[1] for r in c.execute(f"pragma table_info({t})")]
            col = "kernel_name" if "kernel_name" in cols else ("display_name" if "display_name" in cols else None)
            if col:
                for (n,) in c.execute(f"select distinct {col} from {t} where {col} like '%Cijk%'"):
                    print("FULL NAME:", n)
            break
PY
rm -rf $O/p_*
cat $O/summary.txt
