#!/usr/bin/env bash
# Static ablations of conv_d3h_kernel (timing only: results are wrong): one object per -DD3H_NO_* variant, linked into a variant
# library under gpurun_lib/.  Build here:  bash tools/d3h_abl.sh ; on the GPU box:
#   for v in NONE ...; do AB_D3H=1 DDPM_OOD_HIP_LIB=$PWD/gpurun_lib/libd3h_$v.so python tools/wino_ab.py 1024; done
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
mkdir -p "${root}/gpurun_lib"
objs=$(ls "${root}"/build/obj/*.o | grep -v conv_d3h.o)
for v in ${@:-NONE D3H_NO_XLOAD D3H_NO_XSTORE D3H_NO_DMA D3H_NO_MFMA}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -D${v} -c "${root}/ddpm_ood_amd/csrc/conv_d3h.hip" -o "${root}/gpurun_lib/d3h_${v}.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared ${objs} "${root}/gpurun_lib/d3h_${v}.o" -o "${root}/gpurun_lib/libd3h_${v}.so"
done
