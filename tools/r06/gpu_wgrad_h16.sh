# split-f16 weight gradient: tests, then the A/B of the native training step (DDPM_WGRAD_F16X3=0 is the fp32-MFMA form)
o=$GRAFT_REPO_ROOT/gpurun_out/r06_wgrad_h16
mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_train_ops.py -q -x -k "wgrad" 2>&1 | tail -15 | tee $o/tests_wgrad.log
for v in 1 0 1 0; do DDPM_WGRAD_F16X3=$v python tools/train_step_bench.py 64 10 native 2>&1 | grep "images/s" | sed "s/^/F16X3=$v B=64 /"; done | tee $o/wgrad_f16x3_ab.log
for v in 1 0; do DDPM_WGRAD_F16X3=$v python tools/train_step_bench.py 256 10 native 2>&1 | grep "images/s" | sed "s/^/F16X3=$v B=256 /"; done | tee -a $o/wgrad_f16x3_ab.log
timeout 900 python -m pytest tests/test_gpu_train.py -q -x 2>&1 | tail -5 | tee $o/tests_train.log
