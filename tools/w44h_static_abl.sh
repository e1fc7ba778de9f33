#!/bin/bash
# static ablations of conv_wino44h.hip: rebuilds the library per variant (run from the repo root, HERE), then times on the GPU
# box with:  for v in base NO_DMA ...; do DDPM_OOD_HIP_LIB=gpurun_lib/lib_$v.so python tools/wino_ab.py 256; done
mkdir -p gpurun_lib
for v in "$@"; do
  if [ "$v" = base ]; then bash ddpm_ood_amd/csrc/build.sh >/dev/null 2>&1; else bash ddpm_ood_amd/csrc/build.sh -DW44H_$v >/dev/null 2>&1; fi
  cp ddpm_ood_amd/libddpm_ood_hip.so gpurun_lib/lib_$v.so
  echo built $v
done
bash ddpm_ood_amd/csrc/build.sh >/dev/null 2>&1
