mkdir -p gpurun_out/r06/oneiter
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06/oneiter -o c4 -- python3 $GRAFT_REPO_ROOT/tools/c4_iter.py 1000000 1 15 > /dev/null 2>&1
grep "step_kernel\|traj_rows\|compact_list" $(find $GRAFT_REPO_ROOT/gpurun_out/r06/oneiter -name "*kernel_stats.csv") | cut -d, -f2-4
