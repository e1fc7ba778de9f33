# kernel stats of the native training step at batch $1 (default 256) -> gpurun_out/r06_wgrad_h16/
B=${1:-256}
o=$GRAFT_REPO_ROOT/gpurun_out/r06_wgrad_h16
mkdir -p $o
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_train
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -- python $GRAFT_REPO_ROOT/tools/train_step_bench.py $B 5 native > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/prof_train -name "*_results.db" | head -1) > $o/train_native_b${B}_kernel_stats.csv
head -24 $o/train_native_b${B}_kernel_stats.csv | awk -F, '{printf "%-80s %6s %12s %10s %6s\n", substr($1,1,80), $2, $3, $4, $7}'
