#!/bin/bash
# rocprofv3 counter passes over tools/microbench.py (run on the GPU box; one pass per counter group).
#   bash tools/pmc_collect.sh gpurun_out/pmc [microbench args...]
#   PMC_CMD="python tools/vqvae_bench.py 1" bash tools/pmc_collect.sh gpurun_out/pmc_vqvae      # another workload
set -u
out=$1; shift
export TMPDIR=/tmp
mkdir -p "$out"
run() {  # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -o pmc -- ${PMC_CMD:-python tools/microbench.py --iters 2} "${EXTRA[@]}" > "$out/$name.log" 2>&1
  f=$(find "$out/$name" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && mv "$f" "$out/$name/pmc_counter_collection.csv"
  find "$out/$name" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} + 2>/dev/null
}
EXTRA=("$@")
run SQ_WAVE_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU
run SQ_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM
run GRBM_GUI_ACTIVE GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum
run FETCH_SIZE FETCH_SIZE
run WRITE_SIZE WRITE_SIZE
ls -la "$out"/*/
