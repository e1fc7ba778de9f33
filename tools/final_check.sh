#!/bin/bash
# Short end-of-session check on the GPU box: the whole -m gpu suite, smoke, the default bench line and cfg4.
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee gpurun_out/gpu_tests_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/smoke_final.log
python bench.py --steps 2 --warmup 1 > gpurun_out/bench_cfg2.json 2> gpurun_out/bench_cfg2.err
python bench.py --config cfg4 --steps 1 --warmup 1 > gpurun_out/bench_cfg4.json 2> gpurun_out/bench_cfg4.err
for c in 2 4; do python -c "
import json; d=json.load(open('gpurun_out/bench_cfg$c.json')); print('cfg$c', d['value'], d.get('value_batch256'), d['roofline']['profile_key'], d['roofline']['frac'], d['roofline'].get('traffic'))"; done
