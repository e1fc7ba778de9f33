#!/usr/bin/env bash
# Static ablations of attention_fa_kernel (timing only: results are wrong): rebuilds ONE object with -DFA_NO_* and links a variant
# library under gpurun_lib/, then times tools/attn_ab.py against each.  Run on the GPU box: bash tools/fa_abl.sh
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
mkdir -p "${root}/gpurun_lib"
objs=$(ls "${root}"/build/obj/*.o | grep -v attention_fa.o)
for v in NONE FA_NO_DMA FA_NO_SOFTMAX FA_NO_S FA_NO_PV; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -D${v} -c "${root}/ddpm_ood_amd/csrc/attention_fa.hip" -o "${root}/gpurun_lib/afa_${v}.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared ${objs} "${root}/gpurun_lib/afa_${v}.o" -o "${root}/gpurun_lib/lib_${v}.so"
done
if [ "${1:-}" = "run" ]; then
  for v in NONE FA_NO_DMA FA_NO_SOFTMAX FA_NO_S FA_NO_PV; do
    echo "== ${v}"
    DDPM_OOD_HIP_LIB="${root}/gpurun_lib/lib_${v}.so" python "${root}/tools/attn_ab.py" 2>&1 | grep "new" | head -4
  done
fi
