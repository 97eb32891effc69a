timeout 2400 python -m pytest tests/test_gpu_dense_parity.py tests/test_gpu_fullsize.py tests/test_gpu_stationary.py tests/test_gpu_determinism.py tests/test_gpu_state_ops.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
( timeout 300 python tools/fuzz_dense.py 60 5 ) 2>&1 | tail -1
