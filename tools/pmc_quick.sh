#!/bin/bash
# usage: tools/pmc_quick.sh OUTDIR WORKLOAD "CTR1 CTR2 ..." ["CTR ..." ...]   one rocprofv3 --pmc pass per counter group
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/$1; W=$2; shift 2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/g$i" -o $W -- python3 "$ROOT/bench.py" --workload $W --no-cpu-baseline > /dev/null 2> "$OUT/g$i.err"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + '/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:230]
        a = agg[k][r['Counter_Name']]
        a[0] += float(r['Counter_Value']); a[1] += 1
with open(out + '/summary.txt', 'w') as fo:
    for k, d in agg.items():
        tot = sum(v[1] for v in d.values())
        if tot < 20: continue
        fo.write(k + '\n')
        for c, (s, n) in sorted(d.items()):
            fo.write('   %-32s %14.6g  (avg over %d dispatches)\n' % (c, s / n, n))
print(open(out + '/summary.txt').read())
PY
find "$OUT" -name "*.csv" -size +4M -delete
