#!/usr/bin/env bash
# Build libddpm_ood_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${here}/../libddpm_ood_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"${HIPCC}" --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value \
  "${here}/api.hip" "${here}/conv_mfma.hip" "${here}/conv_wino.hip" "${here}/conv1x1_dma.hip" "${here}/conv_direct.hip" "${here}/conv3d_edge.hip" "${here}/groupnorm.hip" \
  "${here}/attention.hip" "${here}/elementwise.hip" "${here}/lpips.hip" "${here}/vq.hip" "${here}/unet_engine.hip" \
  -o "${out}" "$@"
echo "built ${out}"
