# the 64-lane float64 sums on the matrix pipe (wave_sum64): full GPU suite, then C2 against the ladder's library
( time timeout 3000 python -m pytest tests -m gpu -x -q ) > gpurun_out/mfsum_pytest.log 2>&1
grep -E "passed|failed" gpurun_out/mfsum_pytest.log | tail -2
for rep in 1 2 3; do
for v in nomf mjhmc_hip; do
MJHMC_HIP_LIB=$PWD/mjhmc_amd/lib/lib$v.so timeout 900 python bench.py --workload c2 --no-cpu-baseline --shard-of 8 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$v c2_ms', c['c2_ms'], 'one_iter', c.get('c2_one_iter_ms'), 'shard8_ms', c['c2_shard8_ms'], 'eff', c['c2_shard8_eff'])"
done
done
