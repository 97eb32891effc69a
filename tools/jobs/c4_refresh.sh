timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for i in 1 2 3; do timeout 120 python tools/c4_iter.py 1000000 20 15; done
echo NO_RELAY; MJHMC_HIP_LIB=mjhmc_amd/lib/libmjhmc_hip_test.so MJHMC_NO_RELAY=1 timeout 120 python tools/c4_iter.py 1000000 20 15
( timeout 200 python tools/fuzz_rows.py 100 31 ) 2>&1 | tail -1
