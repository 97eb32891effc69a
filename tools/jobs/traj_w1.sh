cd /tmp && export TMPDIR=/tmp
for lib in libmjhmc_hip.so libtraj_w1.so; do
  rm -rf /tmp/kt; MJHMC_HIP_LIB=$GRAFT_REPO_ROOT/mjhmc_amd/lib/$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o c4 -- python3 $GRAFT_REPO_ROOT/tools/c4_iter.py 1000000 1 15 > /dev/null 2>&1
  echo $lib; python3 - <<'PY'
import csv,glob
for f in glob.glob('/tmp/kt/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if any(k in row['Name'] for k in ('traj_rows','step_kernel','compact_list')): print('  ',row['Name'][:40],row['Calls'],row['AverageNs'])
PY
done
