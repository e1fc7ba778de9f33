#!/bin/bash
# Static ablations / variants of conv_wino44r.hip (timing only unless the variant says otherwise): one object per -D set, linked
# with the other objects of the last full build into abl_lib/lib_<tag>.so (git-ignored, travels with gpurun).
# The wrong-result timing ablations (-DW44R_NO_PROD / _NO_PIXEL / _NO_MFMA / _NO_EPI / _NO_ALOAD / _NO_PREAD / _NO_VSTORE /
# _PIX_NOMATH / _PIX_NOLOAD / _PIX_NOSTORE / _PIX_HITLOAD) no longer live in the product source (round 6): they are
# tools/patches/w44r_timing_ablations.patch, applied here to a scratch copy of the kernel before it is compiled.
#   bash tools/w44r_abl.sh base "NO_PROD:-DW44R_NO_PROD" "NO_PIX:-DW44R_NO_PIXEL" ...
# On the GPU box:  for t in base NO_PROD ...; do DDPM_OOD_HIP_LIB=$PWD/abl_lib/lib_$t.so python tools/wino_ab.py 1024; done
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "${root}/abl_lib"
objs=$(ls "${root}"/build/obj/*.o | grep -v conv_wino44r.o)
scratch="$(mktemp -d)"
mkdir -p "${scratch}/ddpm_ood_amd/csrc" "${scratch}/include"
cp "${root}"/ddpm_ood_amd/csrc/*.h "${root}/ddpm_ood_amd/csrc/conv_wino44r.hip" "${scratch}/ddpm_ood_amd/csrc/"
cp "${root}"/include/*.h "${scratch}/include/"
( cd "${scratch}" && patch -p1 -s < "${root}/tools/patches/w44r_timing_ablations.patch" )
pids=()
for spec in "$@"; do
  tag="${spec%%:*}"; flags=""
  if [ "$spec" != "$tag" ]; then flags="${spec#*:}"; fi
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -fno-slp-vectorize ${flags} \
      -c "${scratch}/ddpm_ood_amd/csrc/conv_wino44r.hip" -o "${root}/abl_lib/w44r_${tag}.o" 2>/dev/null && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared ${objs} "${root}/abl_lib/w44r_${tag}.o" -o "${root}/abl_lib/lib_${tag}.so" && \
    rm -f "${root}/abl_lib/w44r_${tag}.o" && echo "built ${tag}" ) &
  pids+=($!)
  if (( ${#pids[@]} >= 3 )); then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
done
for p in "${pids[@]}"; do wait "$p"; done
