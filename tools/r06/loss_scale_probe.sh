# gradient error of the native step against float64 autograd (tests/dev_native_grad_debug.py) for the F(4x4) input-gradient form
# under a power-of-two loss scale
o=$GRAFT_REPO_ROOT/gpurun_out/r06_loss_scale
mkdir -p $o
for dg in wino44h wino; do for sc in 1 1024 65536 4194304; do
  DDPM_TRAIN_DGRAD=$dg DDPM_TRAIN_LOSS_SCALE=$sc python tests/dev_native_grad_debug.py ${1:-4} 2>&1 | grep -v amdgpu > $o/grad_error_${dg}_scale${sc}.log
  echo "dgrad=$dg scale=$sc: worst $(sort -k2 -g -r $o/grad_error_${dg}_scale${sc}.log | grep -v "^loss\|^bad" | head -1)  $(grep '^bad' $o/grad_error_${dg}_scale${sc}.log)"
done; done | tee $o/summary.log
