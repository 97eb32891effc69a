# A/B: the prior's force between the owners' requests (pb1) against behind them (pb0); three alternating runs each
for rep in 1 2 3; do
for v in pb0 pb1; do
  MJHMC_HIP_LIB=$PWD/mjhmc_amd/lib/libsic_$v.so timeout 600 python bench.py --workload c5 --no-cpu-baseline --shard-of 1 --steps 16 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v c5', d['config']['c5_ms'])"
done
done
MJHMC_HIP_LIB=$PWD/mjhmc_amd/lib/libsic_pb1.so timeout 900 python -m pytest tests/test_gpu_dense_parity.py -x -q -k "sic" 2>&1 | grep -E "passed|failed|Error" | tail -2
