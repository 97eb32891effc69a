# the cold list carried between calls: the new test, the fused / parity suites, then C4 one iteration per call with and without
timeout 900 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -3
T=$PWD/mjhmc_amd/lib/libmjhmc_hip_test.so
for rep in 1 2 3; do
MJHMC_HIP_LIB=$T MJHMC_NO_LIST_CARRY=1 timeout 900 python bench.py --workload c4 --no-cpu-baseline --shard-of 1 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('scan  c4_ms', c['c4_ms'], 'one_iter', c.get('c4_one_iter_ms'), c.get('c4_one_iter_frac'))"
MJHMC_HIP_LIB=$T timeout 900 python bench.py --workload c4 --no-cpu-baseline --shard-of 1 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('carry c4_ms', c['c4_ms'], 'one_iter', c.get('c4_one_iter_ms'), c.get('c4_one_iter_frac'))"
done
( time timeout 3000 python -m pytest tests -m gpu -x -q ) > gpurun_out/carry_pytest.log 2>&1
grep -E "passed|failed" gpurun_out/carry_pytest.log | tail -2
