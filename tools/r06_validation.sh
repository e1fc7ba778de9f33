#!/bin/bash
# Round-6 validation on the GPU box (one gpurun call, ~45 min): GPU tests with durations, smoke, the bench lines of every config,
# the rocprofv3 kernel trace of the default bench, PMC passes at the timed batch, the training-step bench + trace, the A/B tools.
# Outputs under gpurun_out/r06/.      gpurun --timeout 3500 -- 'bash tools/r06_validation.sh'
export TMPDIR=/tmp
o=gpurun_out/r06; mkdir -p $o
python -m pytest tests -m gpu -q --durations=15 2>&1 | tail -28 > $o/gpu_tests_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > $o/smoke_final.log
python bench.py --steps 3 --warmup 1 > $o/bench_cfg2.json 2> $o/bench_cfg2.err
python bench.py --config cfg1 --steps 20 --warmup 5 > $o/bench_cfg1.json 2> $o/bench_cfg1.err
python bench.py --config cfg3 --steps 1 --warmup 1 > $o/bench_cfg3.json 2> $o/bench_cfg3.err
python bench.py --config cfg4 --steps 1 --warmup 1 > $o/bench_cfg4.json 2> $o/bench_cfg4.err
python bench.py --config cfg5 --steps 1 --warmup 0 > $o/bench_cfg5.json 2> $o/bench_cfg5.err
python bench.py --batch 128 --steps 1 --warmup 1 --no-cpu-baseline --no-batch256 --no-fp32-products > $o/bench_cfg2_b128.json 2> $o/bench_cfg2_b128.err
# kernel trace of the default bench (batch 1 024)
rocprofv3 --kernel-trace --stats -d $o/prof_kt -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-batch256 --no-fp32-products > $o/bench_b1024_under_rocprofv3.json 2> $o/bench_b1024_under_rocprofv3.err
python tools/rocpd_summary.py $(find $o/prof_kt -name '*.db' | head -1) > $o/bench_b1024_kernel_trace_stats.csv 2> $o/rocpd_summary.err
rm -rf $o/prof_kt
# PMC passes over a B = 1 024 forward
bash tools/pmc_collect.sh $o/pmc --batch 1024 > $o/pmc_collect.log 2>&1
cp profiles/pmc_traffic.json $o/pmc_traffic.json
python tools/pmc_summary.py $o/pmc $o/pmc_per_kernel_b1024.csv $o/pmc_traffic.json --merge --batch 1024 > $o/pmc_summary.log 2>&1
rm -rf $o/pmc/*/pmc_counter_collection.csv
# training step (row f-3): native vs ATen, kernel trace of the native step
for b in 32 64 128 256; do python tools/train_step_bench.py $b 10 both 2>&1 | grep "images/s"; done > $o/train_step_native_vs_aten.log
rocprofv3 --kernel-trace --stats -d $o/prof_train -- python tools/train_step_bench.py 64 5 native > /dev/null 2>&1
python tools/rocpd_summary.py $(find $o/prof_train -name '*.db' | head -1) > $o/train_native_b64_kernel_trace_stats.csv 2>> $o/rocpd_summary.err
rm -rf $o/prof_train
for r in 1 2; do python tools/wino_ab.py 1024 2>&1 | grep -v amdgpu.ids; done > $o/wino44r_six_layers_b1024.log
for r in 1 2; do python tools/wino_ab.py 128 2>&1 | grep -v amdgpu.ids; done > $o/wino44r_six_layers_b128.log
python tools/microbench.py --batch 1024 2>&1 | grep -v amdgpu.ids > $o/microbench_small_b1024.log
python tools/microbench.py --batch 256 2>&1 | grep -v amdgpu.ids > $o/microbench_small_b256.log
python tools/microbench.py --batch 128 2>&1 | grep -v amdgpu.ids > $o/microbench_small_b128.log
python tools/vqvae_bench.py 2 2>&1 | grep -v amdgpu.ids > $o/vqvae_bench_final.log
python tools/conv1x1_ab.py --batch 1024 2>&1 | grep -v amdgpu.ids > $o/conv1x1_ab_final.log
python tests/parity_report.py --n 64 --skip 500 2>&1 | grep -v amdgpu.ids | tail -12 > $o/parity_report_final.log
for c in 1 2 3 4 5; do python -c "
import json; d=json.load(open('$o/bench_cfg$c.json')); print('cfg$c', d['value'], d.get('value_batch256'), d.get('value_fp32_products'), d['roofline']['profile_key'], d['roofline']['frac'])"; done
tail -3 $o/gpu_tests_final.log; tail -4 $o/parity_report_final.log; cat $o/train_step_native_vs_aten.log
