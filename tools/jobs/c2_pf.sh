# A/B: the fused jump kernel's touch of the next slot's rows (MJHMC_PF_AHEAD iterations before a slot ends; 0 = off)
for rep in 1 2 3; do
for v in 0 6 v6; do
  MJHMC_HIP_LIB=$PWD/mjhmc_amd/lib/libpf$v.so timeout 600 python bench.py --workload c2 --no-cpu-baseline --shard-of 8 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('pf$v c2_ms', c['c2_ms'], 'shard8_ms', c['c2_shard8_ms'], 'eff', c['c2_shard8_eff'])"
done
done
MJHMC_HIP_LIB=$PWD/mjhmc_amd/lib/libpfv6.so timeout 900 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -2
