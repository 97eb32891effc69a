# SQ counters of the one-iteration path's two launches (C4, --steps 1): is the jump-process launch bound by its vector instructions?
ROOT=$PWD; OUT=$ROOT/gpurun_out/decide_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/g$i" -o c4 -- python3 "$ROOT/bench.py" --workload c4 --steps 1 --no-cpu-baseline --shard-of 1 > /dev/null 2> "$OUT/g$i.err"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + '/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:90]
        a = agg[k][r['Counter_Name']]
        a[0] += float(r['Counter_Value']); a[1] += 1
with open(out + '/summary.txt', 'w') as fo:
    for k, d in agg.items():
        tot = sum(v[1] for v in d.values())
        if tot < 200: continue
        fo.write(k + '\n')
        for c, (s, n) in sorted(d.items()):
            fo.write('   %-32s %14.6g  (avg over %d dispatches)\n' % (c, s / n, n))
print(open(out + '/summary.txt').read())
PY
find "$OUT" -name "*.csv" -size +1M -delete
