o=$GRAFT_REPO_ROOT/gpurun_out/r06_prof256
mkdir -p $o
cd /tmp && export TMPDIR=/tmp
DDPM_TRAIN_DGRAD=${DG:-wino} timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -- python $GRAFT_REPO_ROOT/tools/train_step_bench.py 256 5 native > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/prof_train -name "*_results.db" | head -1) > $o/train_native_b256_kernel_stats.csv; head -32 $o/train_native_b256_kernel_stats.csv | cut -c1-120,121-180 | awk -F, '{printf "%-90s %6s %12s %10s %6s\n", substr($1,1,90), $2, $3, $4, $7}'
