cd /tmp; export TMPDIR=/tmp
W=${1:-cfg2}
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/qp -o q -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/qp.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/qp -name "*.db" | head -1); python tools/rocprof_summary.py $f > gpurun_out/qp_stats.txt 2>&1; find gpurun_out/qp -name "*.db" -delete
