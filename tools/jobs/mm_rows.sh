timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_stationary.py -x -q 2>&1 | tail -4
echo MM rows relay; for i in 1 2; do timeout 120 python tools/c4_iter.py 1000000 20 15 0.1 mm; done
echo MM group form; MJHMC_HIP_LIB=mjhmc_amd/lib/libmjhmc_hip_test.so MJHMC_NO_ROWS=1 timeout 120 python tools/c4_iter.py 1000000 20 15 0.1 mm
echo MM rows one-wave; MJHMC_HIP_LIB=mjhmc_amd/lib/libmjhmc_hip_test.so MJHMC_NO_RELAY=1 timeout 120 python tools/c4_iter.py 1000000 20 15 0.1 mm
( timeout 300 python tools/fuzz_rows.py 150 11 ) 2>&1 | tail -2
