#!/bin/bash
# Round-3 evidence run on the GPU box (outputs under gpurun_out/; copy the summaries into profiles/):
#   kernel trace of the default bench (batch 1 024), PMC passes over a B = 1 024 forward, microbenchmark breakdowns.
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-batch256 > gpurun_out/r03_bench_b1024_under_rocprofv3.json 2> gpurun_out/r03_bench_b1024_under_rocprofv3.err
python tools/rocpd_summary.py $(find gpurun_out/prof_kt -name '*.db' | head -1) > gpurun_out/r03_bench_b1024_kernel_trace_stats.csv 2> gpurun_out/rocpd_summary.err
rm -rf gpurun_out/prof_kt
python tools/microbench.py --batch 1024 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_microbench_small_b1024.log
python tools/microbench.py --batch 256 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_microbench_small_b256.log
bash tools/pmc_collect.sh gpurun_out/pmc --batch 1024 > gpurun_out/pmc_collect.log 2>&1
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
python tools/pmc_summary.py gpurun_out/pmc gpurun_out/r03_pmc_per_kernel_b1024.csv gpurun_out/pmc_traffic.json --merge --batch 1024 > gpurun_out/pmc_summary.log 2>&1
rm -rf gpurun_out/pmc/*/pmc_counter_collection.csv
tail -5 gpurun_out/pmc_summary.log; head -8 gpurun_out/r03_bench_b1024_kernel_trace_stats.csv | cut -c1-150
