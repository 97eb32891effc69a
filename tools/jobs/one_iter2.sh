timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -k "compact or c4 or funnel or fused or mm" 2>&1 | tail -3
for i in 1 2 3; do timeout 120 python tools/c4_iter.py 1000000 1 15; done
