timeout 900 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -2
for i in 1 2 3; do timeout 120 python tools/c4_iter.py 1000000 20 15; done
echo NO_RELAY; MJHMC_HIP_LIB=mjhmc_amd/lib/libmjhmc_hip_test.so MJHMC_NO_RELAY=1 timeout 120 python tools/c4_iter.py 1000000 20 15
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/pm -o c4 -- python3 $GRAFT_REPO_ROOT/tools/c4_iter.py 1000000 20 15 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('/tmp/pm/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'relay' in row['Kernel_Name']: acc[row['Counter_Name']].append(float(row['Counter_Value']))
for k,v in acc.items(): print(k,'%.4g'%(sum(v)/len(v)))
PY
