# training kernels: tests, step throughput, per-shape profile
o=$GRAFT_REPO_ROOT/gpurun_out/r06_train2
mkdir -p $o
timeout 1200 python -m pytest tests/test_gpu_train_ops.py -q -x 2>&1 | tail -8 | tee $o/tests_train_ops.log
for b in 64 256; do python tools/train_step_bench.py $b 10 native 2>&1 | grep "images/s"; done | tee $o/train_step.log
DDPM_PROF_SHAPES=1 python tools/train_step_bench.py 256 3 native 2>&1 | grep -v amdgpu > $o/train_native_b256_shapes.txt
head -50 $o/train_native_b256_shapes.txt
timeout 900 python -m pytest tests/test_gpu_train.py -q -x 2>&1 | tail -3 | tee $o/tests_train.log
