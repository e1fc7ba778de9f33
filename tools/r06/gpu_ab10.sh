o=$GRAFT_REPO_ROOT/gpurun_out/r06_ab10
mkdir -p $o
for lib in default nomfma nosplit nodma noepi depth1; do
  if [ $lib = default ]; then unset DDPM_OOD_HIP_LIB; else export DDPM_OOD_HIP_LIB=$PWD/abl_lib/lib_c1x1_$lib.so; fi
  echo "== $lib" >> $o/conv1x1_ablations.log
  python tools/conv1x1_ab.py --batch 1024 --child 2>&1 | grep -v amdgpu | grep -v "GN +" >> $o/conv1x1_ablations.log
done
unset DDPM_OOD_HIP_LIB
grep -E "==|sum" $o/conv1x1_ablations.log
for v in 0 1 0 1; do DDPM_WGRAD_OCC2=$v python tools/train_step_bench.py 64 10 native 2>&1 | grep "images/s" | sed "s/^/OCC2=$v /"; done | tee $o/wgrad_occ2_ab.log
for v in 0 1; do DDPM_WGRAD_OCC2=$v python tools/train_step_bench.py 256 10 native 2>&1 | grep "images/s" | sed "s/^/OCC2=$v /"; done | tee -a $o/wgrad_occ2_ab.log
timeout 600 python -m pytest tests/test_gpu_train_ops.py -q -k wgrad 2>&1 | tail -2
DDPM_WGRAD_OCC2=1 timeout 600 python -m pytest tests/test_gpu_train_ops.py -q -k wgrad 2>&1 | tail -2
