set -x
o=gpurun_out/r06_w44r_ir
mkdir -p $o
for rep in 1 2; do
for lib in default abl_lib/libw44r_ir_sets2.so abl_lib/libw44r_ir_sets3.so; do
  if [ $lib = default ]; then unset DDPM_OOD_HIP_LIB; else export DDPM_OOD_HIP_LIB=$PWD/$lib; fi
  echo "== $lib" >> $o/ab_b1024.log
  python tools/wino_ab.py 1024 2>&1 | grep -v amdgpu.ids >> $o/ab_b1024.log
done
done
export DDPM_OOD_HIP_LIB=$PWD/abl_lib/libw44r_ir_sets3.so
timeout 900 python -m pytest tests/test_gpu_wino44h.py -x -q -k "register_fed" > $o/sets3_bit_identity.log 2>&1
tail -3 $o/sets3_bit_identity.log
cat $o/ab_b1024.log | grep -E "==|total"
