#!/usr/bin/env bash
# Build libddpm_ood_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
# One object per source (compiled in parallel, JOBS at a time), then one link; extra arguments go to every compile.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${here}/../libddpm_ood_hip.so"
obj="${here}/../../build/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
JOBS="${JOBS:-4}"
mkdir -p "${obj}"
srcs=(api conv_mfma conv_wino conv_wino44 conv_wino44h conv_wino44r conv_d3s conv_s2h conv1x1_dma conv_direct conv3d_edge linear_skinny groupnorm attention attention_fa elementwise lpips vq
      unet_engine train_gemm train_ops)
common=(--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value)
# per-file flags: declare an array flags_<source> to add options to one translation unit, e.g.
#   flags_attention=(-mllvm -amdgpu-mfma-vgpr-form=1)   # measured: 80 vs 86 TFLOP/s at n = 4096, not used
# conv_wino44h: the SLP vectoriser packs the fp32 transform arithmetic into v_pk_fma_f32 / v_pk_add_f32, which cost more
# than the scalar forms beside MFMAs (MI355X_MICROARCH.md, price of a filler)
flags_conv_wino44h=(-fno-slp-vectorize)
flags_conv_wino44r=(-fno-slp-vectorize)
pids=()
for f in "${srcs[@]}"; do
  extra_name="flags_${f}[@]"
  extra=()
  if declare -p "flags_${f}" >/dev/null 2>&1; then extra=("${!extra_name}"); fi
  "${HIPCC}" "${common[@]}" "${extra[@]}" "$@" -c "${here}/${f}.hip" -o "${obj}/${f}.o" &
  pids+=($!)
  if (( ${#pids[@]} >= JOBS )); then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
done
for p in "${pids[@]}"; do wait "$p"; done
objs=()
for f in "${srcs[@]}"; do objs+=("${obj}/${f}.o"); done
"${HIPCC}" --offload-arch=gfx950 -fPIC -shared "${objs[@]}" -o "${out}"
# every kernel's host stub must be defined in the library itself (a toolchain quirk once dropped them silently: the library
# linked, and only dlopen failed)
if nm -C "${out}" | grep -q " U .*ddpm::"; then
  echo "undefined ddpm:: symbols in ${out}:" >&2
  nm -C "${out}" | grep " U .*ddpm::" | head >&2
  exit 1
fi
echo "built ${out}"
