#!/bin/bash
# LDS counters of the SparseImageCode trajectory kernel (tools/sic_leap_time.py): bank conflicts vs active LDS cycles.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/siclds
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT --output-format csv -d "$OUT/p1" -o p1 -- python3 "$ROOT/tools/sic_leap_time.py" 4 25 > "$OUT/p1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d "$OUT/p2" -o p2 -- python3 "$ROOT/tools/sic_leap_time.py" 4 25 > "$OUT/p2.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/p3" -o p3 -- python3 "$ROOT/tools/sic_leap_time.py" 4 25 > "$OUT/p3.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'sic_leap' in r['Kernel_Name']:
            a = agg[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, (s, n) in sorted(agg.items()):
    print('%-28s %14.6g (avg over %d dispatches)' % (k, s / n, n))
PY
