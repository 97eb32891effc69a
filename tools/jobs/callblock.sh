# the call block (one fill + one copy per mjhmc_iterate call): full GPU suite, then the workloads a call's constant shows in
( time timeout 3000 python -m pytest tests -m gpu -x -q ) > gpurun_out/callblock_pytest.log 2>&1
grep -E "passed|failed" gpurun_out/callblock_pytest.log | tail -2
for rep in 1 2; do
timeout 900 python bench.py --workload c2 --no-cpu-baseline --shard-of 8 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('c2_ms', c['c2_ms'], 'one_iter', c.get('c2_one_iter_ms'), 'shard8_ms', c['c2_shard8_ms'], 'eff', c['c2_shard8_eff'])"
timeout 900 python bench.py --workload c4 --no-cpu-baseline --shard-of 8 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('c4_ms', c['c4_ms'], 'one_iter', c.get('c4_one_iter_ms'), 'shard8_ms', c['c4_shard8_ms'], 'eff', c['c4_shard8_eff'])"
done
