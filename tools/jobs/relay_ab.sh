timeout 600 python -m pytest tests/test_gpu_fused.py -q -k relay 2>&1 | tail -40
