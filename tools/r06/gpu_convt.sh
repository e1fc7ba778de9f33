o=$GRAFT_REPO_ROOT/gpurun_out/r06_convt
mkdir -p $o
timeout 1200 python -m pytest tests/test_gpu_ops.py -q -k "conv_transpose3d" 2>&1 | tail -15 > $o/tests.log; tail -5 $o/tests.log
for v in 0 1 0 1; do DDPM_CONVT_PARITY=$v python tools/vqvae_bench.py 2 2>&1 | grep -v amdgpu.ids | sed "s/^/PARITY=$v /" >> $o/vqvae_bench_ab.log; done
grep "decode_stage_2_outputs\|transpose\|interleave\|wino44h" $o/vqvae_bench_ab.log
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_dispatch.py -q -k "cfg5" 2>&1 | tail -4
