for w in c4 c2; do timeout 600 python bench.py --workload $w --no-cpu-baseline --shard-of 1 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print({k:v for k,v in c.items() if 'one_iter' in k or k.endswith('_ms')})"; done
timeout 600 python tools/call_overhead.py
