#!/bin/bash
# ablation timing of conv_wino44h.hip (results are wrong with a mask; timing only):  bash tools/w44h_abl.sh [batch]
b=${1:-256}
for m in ${MASKS:-0 1 2 4 8 3 7 9 10 12 15}; do echo "== DDPM_W44H_ABL=$m"; DDPM_W44H_ABL=$m python tools/wino_ab.py $b 2>&1 | grep -v amdgpu.ids | awk '{print $1, $2, $3}' | paste -sd' '; done
