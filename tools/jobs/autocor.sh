timeout 900 python -m pytest tests/test_gpu_autocor.py -x -q 2>&1 | tail -5
timeout 300 python bench.py --workload c2 --no-cpu-baseline --shard-of 1 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['autocor_on_device'], d['config']['c2_ms'])"
timeout 300 python bench.py --workload c4 --no-cpu-baseline --shard-of 1 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['autocor_on_device'], d['config']['c4_ms'])"
