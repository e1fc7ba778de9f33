#!/bin/bash
# SQ counters of the Downsample kernels (two rocprofv3 --pmc passes over tools/down_ab.py; run on the GPU box)
export TMPDIR=/tmp
B=${1:-1024}
run() { name=$1; shift; d=gpurun_out/s2h_pmc/$name; mkdir -p $d
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $d -o pmc -- python tools/down_ab.py $B > $d.log 2>&1
  f=$(find $d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && mv "$f" $d/pmc.csv; find $d -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} + 2>/dev/null; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA
python - <<P
import pandas as pd
for n in "ab":
    df=pd.read_csv("gpurun_out/s2h_pmc/%s/pmc.csv"%n)
    df=df[df.Kernel_Name.str.contains("conv_s2h")]
    df["dur"]=(df.End_Timestamp-df.Start_Timestamp)/1e3
    pv=df.pivot_table(index=["Kernel_Name","Dispatch_Id"],columns="Counter_Name",values="Counter_Value").reset_index()
    pv["dur"]=df.groupby("Dispatch_Id").dur.first().reindex(pv.Dispatch_Id).values
    pv["Grid"]=df.groupby("Dispatch_Id").Grid_Size.first().reindex(pv.Dispatch_Id).values
    print(pv.groupby(["Kernel_Name","Grid"]).mean(numeric_only=True).drop(columns=["Dispatch_Id"]).T.to_string())
P
rm -rf gpurun_out/s2h_pmc
