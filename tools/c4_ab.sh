#!/bin/bash
# C4's compacted iteration by the clock (no profiler), a few trajectory lengths, row form and group form
R=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-1000000}
export MJHMC_HIP_LIB=$R/mjhmc_amd/lib/libmjhmc_hip_test.so MJHMC_FUSE_BELOW=0
for rep in 1 2; do
for form in rows groups; do
  if [ $form = groups ]; then export MJHMC_NO_ROWS=1; else unset MJHMC_NO_ROWS; fi
  for L in 1 8 15; do
    echo "$form L=$L N=$N $(timeout 120 python3 $R/tools/c4_iter.py $N 20 $L 2>&1 < /dev/null | tail -1)"
  done
done
done
