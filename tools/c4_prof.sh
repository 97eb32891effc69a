#!/bin/bash
# kernel times of C4's compacted iteration under rocprofv3 for a few trajectory lengths, row form and group form
# usage (on the GPU box): tools/c4_prof.sh [N]
R=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-1000000}
export MJHMC_HIP_LIB=$R/mjhmc_amd/lib/libmjhmc_hip_test.so MJHMC_FUSE_BELOW=0
cd /tmp && export TMPDIR=/tmp
for form in rows groups; do
  if [ $form = groups ]; then export MJHMC_NO_ROWS=1; fi
  for L in 1 8 15; do
    rm -rf /tmp/prof_$form$L
    timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$form$L -o c4 -- python3 $R/tools/c4_iter.py $N 20 $L > /tmp/out_$form$L.txt 2>&1 < /dev/null
    for f in $(find /tmp/prof_$form$L -name "*kernel_stats.csv" < /dev/null); do
      python3 -c "
import csv,sys
for r in list(csv.DictReader(open('$f')))[:2]: print('$form L=$L N=$N', r['Name'][:48], 'calls', r['Calls'], 'avg_us %.1f' % (float(r['AverageNs'])/1e3))"
    done
  done
done
