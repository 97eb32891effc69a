timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_fullsize.py -x -q -k "fused or relay or c4 or funnel" 2>&1 | tail -3
( timeout 200 python tools/fuzz_rows.py 120 13 ) 2>&1 | tail -2
