R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in 0 1; do
  export MJHMC_HIP_LIB=$R/mjhmc_amd/lib/libfun_$v.so
  python bench.py --workload c4 --head c4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d[\"roofline\"]; dev=d.get('device',{}); print('exp_lane0=$v', round(d[\"ms_per_step\"],5), round(r[\"frac\"],4), dev.get('sclk_mhz'), dev.get('power_w'))"
done
done
