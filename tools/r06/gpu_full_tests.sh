o=$GRAFT_REPO_ROOT/gpurun_out/r06_full
mkdir -p $o
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $o/gpu_tests.log; tail -12 $o/gpu_tests.log
