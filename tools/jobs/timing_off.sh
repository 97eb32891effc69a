timeout 900 python -m pytest tests/test_gpu_fused.py -x -q -k "timing or carried" 2>&1 | tail -2
for w in c4 c2; do timeout 600 python bench.py --workload $w --no-cpu-baseline --shard-of 1 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print({k:v for k,v in c.items() if 'one_iter' in k or k.endswith('_ms')})"; done
( time timeout 3000 python -m pytest tests -m gpu -x -q ) > gpurun_out/timing_pytest.log 2>&1
grep -E "passed|failed" gpurun_out/timing_pytest.log | tail -2
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -1
