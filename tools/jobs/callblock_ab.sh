# same-box A/B: HEAD's library (libold) against the call block (libnew)
for rep in 1 2 3; do
for v in old new; do
MJHMC_HIP_LIB=$PWD/mjhmc_amd/lib/lib$v.so timeout 900 python bench.py --workload c2 --no-cpu-baseline --shard-of 8 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$v c2_ms', c['c2_ms'], 'one_iter', c.get('c2_one_iter_ms'), 'shard8_ms', c['c2_shard8_ms'], 'eff', c['c2_shard8_eff'])"
MJHMC_HIP_LIB=$PWD/mjhmc_amd/lib/lib$v.so timeout 900 python bench.py --workload c4 --no-cpu-baseline --shard-of 8 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$v c4_ms', c['c4_ms'], 'one_iter', c.get('c4_one_iter_ms'), 'shard8_ms', c['c4_shard8_ms'], 'eff', c['c4_shard8_eff'])"
done
done
