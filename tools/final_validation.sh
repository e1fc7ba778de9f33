#!/bin/bash
# End-of-session validation on the GPU box (one gpurun call, ~25 min): GPU tests, the bench lines of every config,
# the rocprofv3 kernel trace of the batch-256 bench, per-kernel-class microbenchmarks and the PMC passes.
#   gpurun --timeout 2700 -- 'bash tools/final_validation.sh'      (outputs under gpurun_out/; copy into profiles/)
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/gpu_tests_final.log
python bench.py --steps 2 --warmup 1 > gpurun_out/bench_cfg2.json 2> gpurun_out/bench_cfg2.err
python bench.py --config cfg4 --steps 1 --warmup 1 > gpurun_out/bench_cfg4.json 2> gpurun_out/bench_cfg4.err
python bench.py --config cfg3 --steps 1 --warmup 1 > gpurun_out/bench_cfg3.json 2> gpurun_out/bench_cfg3.err
python bench.py --config cfg5 --steps 1 --warmup 0 > gpurun_out/bench_cfg5.json 2> gpurun_out/bench_cfg5.err
python bench.py --batch 256 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_cfg2_b256.json 2> gpurun_out/bench_cfg2_b256.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -- python bench.py --batch 256 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_under_rocprofv3.json 2> gpurun_out/bench_under_rocprofv3.err
python tools/rocpd_summary.py $(find gpurun_out/prof_kt -name '*.db' | head -1) > gpurun_out/kernel_trace_stats.csv 2> gpurun_out/rocpd_summary.err
rm -rf gpurun_out/prof_kt
python tools/microbench.py --batch 256 2>&1 | grep -v amdgpu.ids > gpurun_out/microbench_small_b256.log
python tools/microbench.py --model big --size 64 --channels 3 --batch 16 2>&1 | grep -v amdgpu.ids > gpurun_out/microbench_big_b16.log
bash tools/pmc_collect.sh gpurun_out/pmc > gpurun_out/pmc_collect.log 2>&1
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
python tools/pmc_summary.py gpurun_out/pmc gpurun_out/pmc_per_kernel.csv gpurun_out/pmc_traffic.json --merge > gpurun_out/pmc_summary.log 2>&1
rm -rf gpurun_out/pmc/*/pmc_counter_collection.csv
tail -3 gpurun_out/bench_cfg5.err; tail -5 gpurun_out/pmc_summary.log
