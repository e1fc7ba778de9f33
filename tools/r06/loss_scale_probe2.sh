# the default step (F(4x4) input gradients under the adaptive scale) against float64 autograd, and the fp32-pipe form beside it
o=$GRAFT_REPO_ROOT/gpurun_out/r06_loss_scale
mkdir -p $o
for b in 4 32; do for dg in default wino; do
  if [ $dg = default ]; then unset DDPM_TRAIN_DGRAD; else export DDPM_TRAIN_DGRAD=$dg; fi
  python tests/dev_native_grad_debug.py $b 2>&1 | grep -v amdgpu > $o/grad_error_${dg}_b$b.log
  echo "batch $b dgrad=$dg: worst $(sort -k2 -g -r $o/grad_error_${dg}_b$b.log | grep -v "^loss\|^bad" | head -1)  $(grep '^bad' $o/grad_error_${dg}_b$b.log)"
done; done | tee $o/summary2.log
unset DDPM_TRAIN_DGRAD
