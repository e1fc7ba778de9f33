#!/usr/bin/env bash
# Build conv_wino44r.hip THROUGH LLVM IR so that register allocation of its pinned accumulators is taken away from the compiler:
# the kernel addresses its accumulator tiles a[0:127] by name inside asm statements, and the function attribute
#     "amdgpu-agpr-alloc"="0,0"  (+ "amdgpu-num-vgpr"="64": 128 unified registers for the allocator on gfx90a+)
# -- not reachable from HIP source -- makes every AGPR off limits to the allocator (no AGPR spill slots, no AV-class values in
# a0..a127) while the 128 named AGPRs still count in the kernel descriptor (128 + 128 registers, two waves per SIMD).  What does
# not fit 128 arch VGPRs then spills HONESTLY, to scratch.  tools/asm_spill_report.py prints the spill picture of the result.
#
#   tools/build_w44r_ir.sh <out.so> [extra -D flags for conv_wino44r.hip, e.g. -DW44R_SETS=3]
#
# Every other translation unit is taken from build/obj (run ddpm_ood_amd/csrc/build.sh first).  VERDICT r5 item 1.
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="$1"; shift
llvm=/opt/rocm/lib/llvm/bin
tmp="$(mktemp -d)"
src="${root}/ddpm_ood_amd/csrc/conv_wino44r.hip"
common=(--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -fno-slp-vectorize)
/opt/rocm/bin/hipcc "${common[@]}" "$@" --cuda-device-only -emit-llvm -S "${src}" -o "${tmp}/dev.ll" 2>/dev/null
sed -i 's/"amdgpu-waves-per-eu"="2"/"amdgpu-waves-per-eu"="2" "amdgpu-agpr-alloc"="0,0" "amdgpu-num-vgpr"="64"/' "${tmp}/dev.ll"
grep -q 'amdgpu-agpr-alloc' "${tmp}/dev.ll"
"${llvm}/llc" -mtriple=amdgcn-amd-amdhsa -mcpu=gfx950 -O3 "${tmp}/dev.ll" -o "${tmp}/dev.s"
python3 "${root}/tools/asm_spill_report.py" "${tmp}/dev.s"
"${llvm}/llc" -mtriple=amdgcn-amd-amdhsa -mcpu=gfx950 -O3 -filetype=obj "${tmp}/dev.ll" -o "${tmp}/dev.o"
"${llvm}/lld" -flavor gnu -m elf64_amdgpu --no-undefined -shared "${tmp}/dev.o" -o "${tmp}/dev.out"
"${llvm}/clang-offload-bundler" -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
  -input=/dev/null -input="${tmp}/dev.out" -output="${tmp}/dev.hipfb"
/opt/rocm/bin/hipcc "${common[@]}" "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang "${tmp}/dev.hipfb" -c "${src}" -o "${tmp}/conv_wino44r.o" 2>/dev/null
objs=()
for o in "${root}"/build/obj/*.o; do
  [[ "$(basename "$o")" == conv_wino44r.o ]] || objs+=("$o")
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared "${objs[@]}" "${tmp}/conv_wino44r.o" -o "${out}"
if nm -C "${out}" | grep -q " U .*ddpm::"; then echo "undefined ddpm:: symbols in ${out}" >&2; exit 1; fi
rm -rf "${tmp}"
echo "built ${out}"
