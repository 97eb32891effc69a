R=$PWD
export MJHMC_HIP_LIB=$R/mjhmc_amd/lib/libmjhmc_hip_test.so MJHMC_FUSE_BELOW=0
cd /tmp && export TMPDIR=/tmp
for beta in 0.1 0.000001; do
  rm -rf /tmp/prof_b
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o c4 -- python3 $R/tools/c4_iter.py 1000000 20 15 $beta > /tmp/out_b.txt 2>&1 < /dev/null
  for f in $(find /tmp/prof_b -name "*kernel_stats.csv" < /dev/null); do
    python3 -c "
import csv
for r in list(csv.DictReader(open('$f')))[:2]: print('beta=$beta', r['Name'][:48], 'calls', r['Calls'], 'avg_us %.1f' % (float(r['AverageNs'])/1e3))"
  done
done
