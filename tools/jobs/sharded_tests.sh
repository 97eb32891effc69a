timeout 900 python -m pytest tests/test_gpu_sharded.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
