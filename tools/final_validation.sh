#!/bin/bash
# End-of-session validation on the GPU box (one gpurun call, ~35 min): GPU tests, the bench lines of every config, the rocprofv3
# kernel trace of the default bench, per-kernel-class microbenchmarks and the PMC passes at the timed batch.
#   gpurun --timeout 3300 -- 'bash tools/final_validation.sh'      (outputs under gpurun_out/; copy into profiles/)
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee gpurun_out/gpu_tests_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/smoke_final.log
python bench.py --steps 2 --warmup 1 > gpurun_out/bench_cfg2.json 2> gpurun_out/bench_cfg2.err
python bench.py --config cfg1 --steps 20 --warmup 5 > gpurun_out/bench_cfg1.json 2> gpurun_out/bench_cfg1.err
python bench.py --config cfg3 --steps 1 --warmup 1 > gpurun_out/bench_cfg3.json 2> gpurun_out/bench_cfg3.err
python bench.py --config cfg4 --steps 1 --warmup 1 > gpurun_out/bench_cfg4.json 2> gpurun_out/bench_cfg4.err
python bench.py --config cfg5 --steps 1 --warmup 0 > gpurun_out/bench_cfg5.json 2> gpurun_out/bench_cfg5.err
bash tools/r03_profile.sh > gpurun_out/r03_profile.log 2>&1
python bench.py --batch 128 --steps 1 --warmup 1 --no-cpu-baseline --no-batch256 > gpurun_out/bench_cfg2_b128.json 2> gpurun_out/bench_cfg2_b128.err  # what a rank holds at N = 8 (strong scaling)
DDPM_PROF_SHAPES=1 python tools/microbench.py --batch 1024 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_microbench_small_b1024_per_shape.log
python tools/microbench.py --batch 128 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_microbench_small_b128.log
PMC_CMD="python tools/vqvae_bench.py 1" bash tools/pmc_collect.sh gpurun_out/pmc_vqvae > gpurun_out/pmc_vqvae_collect.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_vqvae gpurun_out/r03_pmc_per_kernel_vqvae.csv gpurun_out/pmc_traffic_vqvae_scratch.json > gpurun_out/pmc_vqvae_summary.log 2>&1
rm -rf gpurun_out/pmc_vqvae/*/pmc_counter_collection.csv
python tools/up_ab.py 1024 2>&1 | grep -v amdgpu.ids > gpurun_out/up_ab_final.log
python tools/wino_ab.py 256 2>&1 | grep -v amdgpu.ids > gpurun_out/wino_ab_final.log
python tools/wino_ab.py 1024 2>&1 | grep -v amdgpu.ids >> gpurun_out/wino_ab_final.log
DDPM_WINO44_F16X3=0 python tools/wino_ab.py 1024 2>&1 | grep -v amdgpu.ids >> gpurun_out/wino_ab_final.log
python tools/vqvae_bench.py 2 2>&1 | grep -v amdgpu.ids > gpurun_out/vqvae_bench_final.log
python tools/attn_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/attn_ab_final.log
python tests/parity_report.py --n 64 --skip 500 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/parity_report_final.log
for c in 1 2 3 4 5; do python -c "
import json; d=json.load(open('gpurun_out/bench_cfg$c.json')); print('cfg$c', d['value'], d.get('value_batch256'), d['roofline']['profile_key'], d['roofline']['frac'])"; done
tail -3 gpurun_out/gpu_tests_final.log; tail -4 gpurun_out/parity_report_final.log
