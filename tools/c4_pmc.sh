#!/bin/bash
# PMC counters of C4's trajectory launch (row form unless MJHMC_NO_ROWS=1), per kernel averaged over the launches
# usage (on the GPU box): [FUSED=1] [ITS=iterations per call] tools/c4_pmc.sh "CTR1 CTR2 ..." [N] [L]
R=$(cd "$(dirname "$0")/.." && pwd)
CTRS=${1:-"SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"}
N=${2:-1000000}
L=${3:-15}
export MJHMC_HIP_LIB=$R/mjhmc_amd/lib/libmjhmc_hip_test.so
if [ -z "${FUSED:-}" ]; then export MJHMC_FUSE_BELOW=0; fi   # FUSED=1: the product's choice (fused launches in row form)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_c4
timeout 200 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d /tmp/pmc_c4 -o c4 -- python3 $R/tools/c4_iter.py $N ${ITS:-5} $L > /tmp/out_pmc.txt 2>&1 < /dev/null
for f in $(find /tmp/pmc_c4 -name "*counter_collection.csv" < /dev/null); do
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Kernel_Name'][:50]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if 'mjhmc' not in k: continue
    print(k, ' '.join('%s=%.4g(n=%d)' % (c, sum(v) / len(v), len(v)) for c, v in sorted(d.items())))
PY
done
