mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_gpu_fused.py -x -q -k relay 2>&1 | tail -5
( timeout 400 python tools/fuzz_rows.py 300 6 ) > gpurun_out/r06/fuzz_rows.txt 2>&1
tail -5 gpurun_out/r06/fuzz_rows.txt
