timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_stationary.py -x -q 2>&1 | tail -2
echo L=5; for i in 1 2; do timeout 120 python tools/c4_iter.py 1000000 20 5; done
echo L=15 NO_RELAY; MJHMC_HIP_LIB=mjhmc_amd/lib/libmjhmc_hip_test.so MJHMC_NO_RELAY=1 timeout 120 python tools/c4_iter.py 1000000 20 15
( timeout 200 python tools/fuzz_rows.py 100 41 ) 2>&1 | tail -1
