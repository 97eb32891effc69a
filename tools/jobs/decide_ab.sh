for rep in 1 2 3 4; do
for v in old new; do
MJHMC_HIP_LIB=$PWD/mjhmc_amd/lib/lib$v.so timeout 900 python bench.py --workload c4 --no-cpu-baseline --shard-of 1 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$v one_iter', c.get('c4_one_iter_ms'), 'call', c.get('c4_one_iter_call_ms'))"
done
done
MJHMC_HIP_LIB=$PWD/mjhmc_amd/lib/libnew.so timeout 900 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -2
