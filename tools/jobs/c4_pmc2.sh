mkdir -p gpurun_out/r06/c4pmc
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/r06/c4pmc/$tag -o c4 -- python3 $R/tools/c4_iter.py 1000000 20 15 > /dev/null 2> $R/gpurun_out/r06/c4pmc/$tag.err
done
python3 - <<'PY'
import csv,glob,os,collections
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r06/c4pmc'
for f in sorted(glob.glob(root+'/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if 'relay' in row['Kernel_Name']:
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
    for k,v in acc.items():
        print(k, 'per launch %.4g' % (sum(v)/len(v)), 'n', len(v))
PY
find $R/gpurun_out/r06/c4pmc -name "*.csv" -size +2M -delete
