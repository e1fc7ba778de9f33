#!/bin/bash
# L2-miss traffic (rocprofv3 FETCH_SIZE, raw KB per launch) of the 3-D split-f16 F(4x4) kernel in one README-VQ-VAE decode, for the
# shipped library and for static ablation builds (tools/w44h_static_abl.sh NO_DMA NO_PIXEL): without the U (weight) LDS-DMA what is
# left are the activation reads, without the pixel loads what is left is the weight stream.  Run on the GPU box from the repo root.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_vq4
for v in "${@:-base}"; do
  lib=""; [ "$v" != base ] && lib="$PWD/gpurun_lib/lib_$v.so"
  for c in FETCH_SIZE; do
    DDPM_OOD_HIP_LIB=$lib timeout 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_vq4/$v -o pmc -- python tools/vqvae_bench.py 1 > gpurun_out/pmc_vq4/$v.log 2>&1
    grep "decode_stage_2_outputs\|^conv3d_wino44h" gpurun_out/pmc_vq4/$v.log | sed "s/^/[$v] /"
    f=$(find gpurun_out/pmc_vq4/$v -name '*counter_collection.csv' | head -1)
    python - "$f" $c $v <<'PY'
import csv, sys, collections
f, c, v = sys.argv[1:4]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != c or "wino44h_kernel" not in r["Kernel_Name"]: continue
    k = r["Kernel_Name"][:70]
    agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
for k, (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"[{v}] {c} {k:70s} launches {n:4d}  KB per launch {s / n:.4g}")
PY
    rm -rf gpurun_out/pmc_vq4/$v
  done
done
