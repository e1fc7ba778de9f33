#!/bin/bash
# Timing ablations of the split-f16 weight-gradient kernel (train_gemm.hip): wrong results by design, built from a scratch copy of
# the source (the product source carries no ablation switches) into abl_lib/lib_wgrad_<tag>.so (git-ignored, travels with gpurun).
#   bash tools/wgrad_abl.sh          # builds base, nomfma, nostage, nofetch
# On the GPU box:  for t in base nomfma nostage nofetch; do DDPM_OOD_HIP_LIB=$PWD/abl_lib/lib_wgrad_$t.so python tools/wgrad_ab.py; done
# (profiles/r06_wgrad_f16x3_variants.log holds the numbers of the kernel's FIRST form; the patterns below follow the kept form and
# fail loudly when the kernel's text moves.)
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "${root}/abl_lib"
objs=$(ls "${root}"/build/obj/*.o | grep -v train_gemm.o)
for tag in base nomfma nostage nofetch; do
  scratch="$(mktemp -d)"
  mkdir -p "${scratch}/ddpm_ood_amd/csrc" "${scratch}/include"
  cp "${root}"/ddpm_ood_amd/csrc/*.h "${root}/ddpm_ood_amd/csrc/train_gemm.hip" "${scratch}/ddpm_ood_amd/csrc/"
  cp "${root}"/include/*.h "${scratch}/include/"
  python3 - "$tag" "${scratch}/ddpm_ood_amd/csrc/train_gemm.hip" <<'PY'
import sys
tag, path = sys.argv[1:3]
s = open(path).read()
def sub(old, new, count=None):
    global s
    assert old in s, f"{tag}: pattern not found: {old[:60]}"
    s = s.replace(old, new) if count is None else s.replace(old, new, count)
if tag == "nomfma":  # the products replaced by one add that keeps every operand alive
    sub("#include <algorithm>\n", "#include <algorithm>\n#undef DDPM_MFMA_F16X3\n#define DDPM_MFMA_F16X3(acc, ah, al, as, bh, bl, bs) do { (acc)[0] += "
        "(float)((ah)[0] + (al)[0] + (as)[0]) + (float)((bh)[0] + (bl)[0] + (bs)[0]); } while (0)\n", 1)
elif tag == "nostage":  # no conversion, no LDS stores (the condition is false at run time, unknown at compile time)
    sub("    if (it >= NA) return;  // (compile time)\n    const int qu = it * qstep, ru = qu >> 6, cu = qu & 63;\n    const float v[4] = {ra[it][0] * sA",
        "    if (it >= NA || p.B > 0) return;\n    const int qu = it * qstep, ru = qu >> 6, cu = qu & 63;\n    const float v[4] = {ra[it][0] * sA")
    sub("  auto stage_d = [&](int it) __attribute__((always_inline)) {\n", "  auto stage_d = [&](int it) __attribute__((always_inline)) {\n    if (p.B > 0) return;\n")
elif tag == "nofetch":  # no global loads (every quad takes the zero of the select)
    sub("    const bool ok = f_zok && yi >= 0 && yi < p.Hi;", "    const bool ok = p.B < 0 && f_zok && yi >= 0 && yi < p.Hi;")
    sub("    const bool ok = d_px0 < f_rows_px;", "    const bool ok = p.B < 0 && d_px0 < f_rows_px;")
    # (the select still issues one load per slot -- of the tensor's first quad, an L2 hit)
open(path, "w").write(s)
PY
  f="${scratch}/ddpm_ood_amd/csrc/train_gemm.hip"
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -c "$f" -o "${root}/abl_lib/train_gemm_${tag}.o" 2>/dev/null && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared ${objs} "${root}/abl_lib/train_gemm_${tag}.o" -o "${root}/abl_lib/lib_wgrad_${tag}.so" && \
    rm -f "${root}/abl_lib/train_gemm_${tag}.o" && echo "built ${tag}" ) &
done
wait
