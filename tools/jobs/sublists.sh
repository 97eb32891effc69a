# the cold lists as 16 sub-lists with a counter line each (one atomic request per workgroup of the jump-process launch)
timeout 900 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -3
for rep in 1 2 3; do
for v in old_test mjhmc_hip_test; do
MJHMC_HIP_LIB=$PWD/mjhmc_amd/lib/lib$v.so timeout 900 python bench.py --workload c4 --no-cpu-baseline --shard-of 1 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$v c4_ms', c['c4_ms'], 'one_iter', c.get('c4_one_iter_ms'), c.get('c4_one_iter_frac'), 'call', c.get('c4_one_iter_call_ms'))"
done
done
( time timeout 3000 python -m pytest tests -m gpu -x -q ) > gpurun_out/sublists_pytest.log 2>&1
grep -E "passed|failed" gpurun_out/sublists_pytest.log | tail -2
