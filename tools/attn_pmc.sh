cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT"; do
  rm -rf gpurun_out/attn_pmc; timeout 200 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/attn_pmc -o pmc -- python tools/attn_ab.py > /dev/null 2>&1
  f=$(find gpurun_out/attn_pmc -name '*counter_collection.csv' | head -1)
  python - "$f" "${KERNEL:-attention}" "${MIN_US:-800}" <<'PY'
import sys, pandas as pd
df = pd.read_csv(sys.argv[1])
df = df[df.Kernel_Name.str.contains(sys.argv[2] if len(sys.argv) > 2 else "attention")]
df["dur_us"] = (df.End_Timestamp - df.Start_Timestamp) / 1e3
big = df[df.dur_us > float(sys.argv[3] if len(sys.argv) > 3 else 800)]   # the N = 4096 launches
pv = big.pivot_table(index="Dispatch_Id", columns="Counter_Name", values="Counter_Value").mean()
print("dur_us", big.groupby("Dispatch_Id").dur_us.first().mean()); print(pv.to_string())
PY
done
rm -rf gpurun_out/attn_pmc
