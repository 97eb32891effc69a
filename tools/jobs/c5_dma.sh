timeout 1500 python -m pytest tests/test_gpu_dense_parity.py -x -q -k "sic" 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 600 python bench.py --workload c5 --no-cpu-baseline --shard-of 1 --steps 16 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5', d['config']['c5_ms'], d['config']['c5_frac'])"
timeout 600 python bench.py --workload c5bf16 --no-cpu-baseline --shard-of 1 --steps 16 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5bf16', d['config']['c5bf16_ms'], d['config']['c5bf16_frac'])"
