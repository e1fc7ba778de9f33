#!/bin/bash
# Timing ablations of the split-f16 weight-gradient kernel (train_gemm.hip): wrong results by design, built from a scratch copy of
# the source (the product source carries no ablation switches) into abl_lib/lib_wgrad_<tag>.so (git-ignored, travels with gpurun).
#   bash tools/wgrad_abl.sh          # builds base, nomfma, nostage, nofetch
# On the GPU box:  for t in base nomfma nostage nofetch; do DDPM_OOD_HIP_LIB=$PWD/abl_lib/lib_wgrad_$t.so python tools/wgrad_ab.py; done
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "${root}/abl_lib"
objs=$(ls "${root}"/build/obj/*.o | grep -v train_gemm.o)
for tag in base nomfma nostage nofetch; do
  scratch="$(mktemp -d)"
  mkdir -p "${scratch}/ddpm_ood_amd/csrc" "${scratch}/include"
  cp "${root}"/ddpm_ood_amd/csrc/*.h "${root}/ddpm_ood_amd/csrc/train_gemm.hip" "${scratch}/ddpm_ood_amd/csrc/"
  cp "${root}"/include/*.h "${scratch}/include/"
  f="${scratch}/ddpm_ood_amd/csrc/train_gemm.hip"
  case $tag in
    nomfma)  # the products replaced by one add that keeps every operand alive
      sed -i 's|^#include <algorithm>|#include <algorithm>\n#undef DDPM_MFMA_F16X3\n#define DDPM_MFMA_F16X3(acc, ah, al, as, bh, bl, bs) do { (acc)[0] += (float)((ah)[0] + (al)[0] + (as)[0]) + (float)((bh)[0] + (bl)[0] + (bs)[0]); } while (0)|' "$f" ;;
    nostage)  # no conversion, no LDS stores (the condition is false at run time, unknown at compile time)
      sed -i 's|if (a_r\[it\] >= 0) {|if (a_r[it] >= 0 \&\& p.B < 0) {|; s|if (d_px\[it\] < (1 << 20)) {|if (d_px[it] < (1 << 20) \&\& p.B < 0) {|' "$f" ;;
    nofetch)  # no global loads
      sed -i 's|ra\[it\] = zok \&\& yi >= 0|ra[it] = p.B < 0 \&\& zok \&\& yi >= 0|; s|rd\[it\] = d_px\[it\] < rows_px|rd[it] = p.B < 0 \&\& d_px[it] < rows_px|' "$f" ;;
  esac
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -c "$f" -o "${root}/abl_lib/train_gemm_${tag}.o" 2>/dev/null && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared ${objs} "${root}/abl_lib/train_gemm_${tag}.o" -o "${root}/abl_lib/lib_wgrad_${tag}.so" && \
    rm -f "${root}/abl_lib/train_gemm_${tag}.o" && echo "built ${tag}" ) &
done
wait
