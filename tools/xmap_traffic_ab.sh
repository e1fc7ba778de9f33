#!/bin/bash
# HBM traffic of the F(4x4) split-f16 kernel with the cout tiles of a slot on ONE XCD (DDPM_WINO44_XMAP=0) against one cout tile
# per XCD (default): FETCH_SIZE / WRITE_SIZE passes over a B = 1 024 forward (run on the GPU box).
export TMPDIR=/tmp
for x in 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    d=gpurun_out/xmap$x/$c; mkdir -p $d
    DDPM_WINO44_XMAP=$x timeout 300 rocprofv3 --pmc $c --output-format csv -d $d -o pmc -- python tools/microbench.py --iters 2 --batch 1024 > $d.log 2>&1
    f=$(find $d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && mv "$f" $d/pmc_counter_collection.csv
    find $d -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} + 2>/dev/null
  done
  python - <<P
import pandas as pd
r={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    df=pd.read_csv("gpurun_out/xmap$x/%s/pmc_counter_collection.csv"%c)
    df=df[df.Kernel_Name.str.contains("conv_wino44h_kernel")]
    df["dur"]=(df.End_Timestamp-df.Start_Timestamp)/1e3
    r[c]=df.groupby("Kernel_Name").agg(n=("Counter_Value","size"),kb=("Counter_Value","mean"),dur=("dur","mean"))
m=r["FETCH_SIZE"].join(r["WRITE_SIZE"],rsuffix="_w")
m["MB"]=(2*m.kb+m.kb_w)*1024/1e6
print("XMAP=$x"); print(m[["n","dur","MB"]].to_string()); print("launch-weighted MB", (m.MB*m.n).sum()/m.n.sum(), "us", (m.dur*m.n).sum()/m.n.sum())
P
  rm -rf gpurun_out/xmap$x
done
