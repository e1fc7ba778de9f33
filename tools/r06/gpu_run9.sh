o=$GRAFT_REPO_ROOT/gpurun_out/r06_run9
mkdir -p $o
python tools/r06/make_w44_digests.py 2>/dev/null > tests/golden/wino44h_digests.json; cp tests/golden/wino44h_digests.json $o/; wc -c $o/wino44h_digests.json
timeout 2400 python -m pytest tests/test_gpu_wino44h.py tests/test_gpu_train_ops.py tests/test_gpu_train.py tests/test_gpu_guard.py -q 2>&1 | tail -30 > $o/t1.log; tail -4 $o/t1.log
timeout 2400 python -m pytest tests/test_gpu_configs.py -q -k "cfg5" -s 2>&1 | tail -60 > $o/cfg5.log; grep -n "cfg5\|passed\|failed\|Error" $o/cfg5.log | tail -20
timeout 2400 python -m pytest tests/test_gpu_dispatch.py -q -k "t990" -s 2>&1 | tail -30 > $o/t990.log; grep -n "cfg4\|passed\|failed\|Error" $o/t990.log | tail
timeout 2400 python -m pytest tests/test_gpu_dist.py -q 2>&1 | tail -15 > $o/dist.log; tail -3 $o/dist.log
for b in 64; do timeout 600 python tools/train_step_bench.py $b 10 both 2>&1 | grep "images/s" | tee -a $o/train_step_bench.log; done
